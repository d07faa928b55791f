"""GPU: config 5 end to end on the 400-station golden database -- the three leave-one-out farms of
topowx_amd.xval (step21 / step23 / step24) over ALL its stations, sampled stations against the oracle, and the
bandwidth optimisation written back into the station table (optimize.py:268-374)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _pt(orc, c, j):
    return orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])


def test_step21_farm_and_optimisation(golden_case, orc):
    from topowx_amd import stationdb as sdb, xval
    _, tmin, _ = golden_case
    stn = sdb.StationSerialDataDb(tmin.stns.copy(), "tmin", tmin.days, tmin.var)
    ids, mae = xval.optim_nstns_norms(stn, "tmin", batch=64)
    good = np.isnan(stn.stns[sdb.BAD])
    assert ids.size == int((np.isfinite(stn.stns[sdb.MASK]) & good).sum()) and mae.shape == (12, 16, ids.size)
    assert np.isfinite(mae).mean() > 0.95                      # rim stations may lack 148 neighbours at k = 147
    db, prm = orc.Db(stn), orc.params()
    c = db.cols
    idx = {s: i for i, s in enumerate(stn.stns[sdb.STN_ID][good])}
    rng = np.random.default_rng(21)
    for q in rng.choice(ids.size, 5, replace=False):
        j = idx[ids[q]]
        for x in (0, 7, 13):
            rc, want, _ = orc.krigall(db, prm, _pt(orc, c, j), int(xval.DFLT_LADDER[x]), excl=j, rm_zero_dist=True)
            if rc:
                assert np.isnan(mae[:, :, q]).all()            # the whole station is abandoned (step21:55-62)
                continue
            assert np.abs(mae[:, x, q] - np.abs(want - c["norm"][:, j])).max() < TOL
    # reduction: every station of a division gets the division's best bandwidth
    chosen = xval.set_optim_nstns_tair_norm(stn, ids, mae)
    div = stn.stns[sdb.CLIMDIV]
    assert len(chosen) == np.unique(div[np.isfinite(div)]).size
    for d, pick in chosen.items():
        cols = np.nonzero(np.isin(ids, stn.stns[sdb.STN_ID][div == d]))[0]
        for m in range(12):
            want = xval.DFLT_LADDER[int(np.argmin(np.nanmean(mae[m][:, cols], axis=1)))]
            assert pick[m] == want and np.all(stn.stns[sdb.get_optim_varname(m + 1)][div == d] == want)


def test_step23_farm(golden_case, golden_xval, orc):
    from topowx_amd import stationdb as sdb, xval
    _, tmin, _ = golden_case
    g = golden_xval
    ids, mae, bias, r2 = xval.optim_nstns_anoms(tmin, "tmin", batch=50)
    assert mae.shape == (12, 16, ids.size) and np.isfinite(mae).mean() > 0.95
    ok = np.isfinite(mae).all(axis=(0, 1))
    assert (mae[:, :, ok] > 0).all() and ((r2[:, :, ok] >= 0) & (r2[:, :, ok] <= 1)).all()
    # the goldens' stations, made by executing the reference's run_xval: [nb, 12] there, month-major here
    good_ids = tmin.stns[sdb.STN_ID][np.isnan(tmin.stns[sdb.BAD])]
    for i, j in enumerate(g["xa_stn"]):
        q = np.nonzero(ids == good_ids[j])[0]
        if q.size == 0:
            continue                                           # a station outside the mask is not cross-validated
        assert np.abs(mae[:, :, q[0]] - g["xa_mae"][i].T).max() < TOL
        assert np.abs(bias[:, :, q[0]] - g["xa_bias"][i].T).max() < TOL
        assert np.abs(r2[:, :, q[0]] - g["xa_r2"][i].T).max() < 1e-6
    chosen = xval.set_optim_nstns_tair_anom(sdb.StationSerialDataDb(tmin.stns.copy(), "tmin", tmin.days), ids, mae)
    assert all(set(v) <= set(xval.DFLT_LADDER) for v in chosen.values())


def test_step24_farm(golden_case, golden, orc):
    from topowx_amd import stationdb as sdb, xval
    _, tmin, _ = golden_case
    ids, norms, se, dly, st = xval.xval_interp(tmin, "tmin", daily=True, batch=128)
    assert norms.shape == (ids.size, 12) and dly.shape == (ids.size, tmin.days.size) and dly.dtype == np.float32
    assert (st == 0).mean() > 0.95 and np.isnan(norms[st != 0]).all() and np.isfinite(norms[st == 0]).all()
    good_ids = tmin.stns[sdb.STN_ID][np.isnan(tmin.stns[sdb.BAD])]
    for i, j in enumerate(golden["xv_idx"]):                   # executed XvalTairOverall.run_interp goldens
        q = np.nonzero(ids == good_ids[j])[0]
        if q.size:
            assert np.abs(norms[q[0]] - golden["xv_norms"][i]).max() < TOL
            assert np.abs(dly[q[0]] - golden["xv_daily"][i]).max() < 1e-3          # float32 storage of degC
    db, prm = orc.Db(tmin), orc.params()
    c = db.cols
    idx = {s: i for i, s in enumerate(good_ids)}
    for q in np.random.default_rng(24).choice(ids.size, 6, replace=False):
        j = idx[ids[q]]
        rc, d, wn, ws = orc.interp(db, prm, _pt(orc, c, j), excl=j, rm_zero_dist=True, daily=True)
        assert rc == st[q]
        if rc == 0:
            assert np.abs(norms[q] - wn).max() < TOL and np.abs(se[q] - ws).max() < TOL
            assert np.abs(dly[q] - d).max() < 1e-3
