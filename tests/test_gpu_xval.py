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


def test_step22_farm(golden_case, orc):
    """step22 (step22:33-142): every cross-validated station's variogram for every month, fitted with its optimised
    bandwidth and written into vario_*MM; sampled stations against the oracle's restatement of get_vario_params
    (BuildKrigParams.get_krig_params: bandwidth smoothed, the station stays in its own neighbourhood)."""
    from topowx_amd import stationdb as sdb, xval
    _, tmin, _ = golden_case
    stn = sdb.StationSerialDataDb(tmin.stns.copy(), "tmin", tmin.days, None)
    before = stn.stns.copy()
    ids, nug, psill, rng = xval.set_stn_variograms(stn, "tmin", batch=100)
    good = np.isnan(before[sdb.BAD])
    assert ids.size == int((np.isfinite(before[sdb.MASK]) & good).sum()) and nug.shape == (12, ids.size)
    ok = np.isfinite(nug).all(axis=0)
    assert ok.mean() > 0.95 and (nug[:, ok] >= 0).all() and (psill[:, ok] >= 0).all() and (rng[:, ok] >= 0).all()
    # a failed station keeps NaN in all three parameters of all twelve months
    assert np.isnan(psill[:, ~ok]).all() and np.isnan(rng[:, ~ok]).all()
    # written into the table for exactly the cross-validated stations; everything else untouched
    pos = {s: i for i, s in enumerate(stn.stns[sdb.STN_ID])}
    rows = np.array([pos[s] for s in ids])
    other = np.setdiff1d(np.arange(stn.stns.size), rows)
    for m in range(1, 13):
        for a, nm in ((nug, sdb.VARIO_NUG), (psill, sdb.VARIO_PSILL), (rng, sdb.VARIO_RNG)):
            f = sdb.get_krigparam_varname(m, nm)
            np.testing.assert_array_equal(stn.stns[f][rows], a[m - 1])
            np.testing.assert_array_equal(stn.stns[f][other], before[f][other])
    # against the oracle, which fits on the table as it was BEFORE the farm wrote into it
    db, prm = orc.Db(sdb.StationSerialDataDb(before, "tmin", tmin.days, None)), orc.params()
    c = db.cols
    idx = {s: i for i, s in enumerate(before[sdb.STN_ID][good])}
    for q in np.random.default_rng(22).choice(ids.size, 6, replace=False):
        j = idx[ids[q]]
        for m in (1, 6, 11):
            rc, v, _ = orc.build_krig_params(db, prm, _pt(orc, c, j), m)
            if rc:
                assert not ok[q]
                continue
            np.testing.assert_allclose([nug[m - 1, q], psill[m - 1, q], rng[m - 1, q]], v, rtol=1e-6, atol=1e-9)


def test_pipeline_step21_22_then_grid_with_fitted_variograms(golden_case, orc):
    """The reference's own order of use: step21 optimises the bandwidths, step22 fits every station's variogram
    (nugget = min gamma: small nuggets occur), and step25 then interpolates the grid with the neighbour-weighted means of
    THOSE parameters.  Grid normals / SE on such a table against the oracle -- whatever share of the systems the small
    fitted nuggets route to the fp64 covariance build."""
    from topowx_amd import _lib, stationdb as sdb, xval
    grid, tmin, _ = golden_case
    stn = sdb.StationDataWrkChk(tmin.stns.copy(), "tmin", tmin.days, None)
    ids, mae = xval.optim_nstns_norms(stn, "tmin", batch=64)
    xval.set_optim_nstns_tair_norm(stn, ids, mae)
    _, nug, psill, rng = xval.set_stn_variograms(stn, "tmin", batch=100)
    ok = np.isfinite(nug)
    small = ok & (rng > 0) & (psill > 0) & (16 * nug < psill)
    assert ok.mean() > 0.95 and small.sum() > 0                  # fitted tables do hold nuggets below psill / 16
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)
    rs, cs = slice(24, 56), slice(30, 70)
    got = ctx.interp_grid(grid, variables=("tmin",), daily=False, rows=rs, cols=cs)
    t = ctx.timing()
    ctx.close()
    want = orc.interp_grid(orc.Db(stn), None, orc.params(), grid, daily=False, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(got["status"], want["status"]) and (want["status"] == 0).mean() > 0.9
    okc = want["status"] == 0
    for k in ("norm_tmin", "se_tmin"):
        assert np.abs(got[k].astype(np.float64) - want[k])[:, okc].max() < TOL, k
    assert t["uk_solves"] == int(okc.sum()) * 12 and 0 <= t["uk_f64_solves"] <= t["uk_solves"]


def test_krigall_points_equals_fit_then_krig(golden_case):
    """twx_krigall_points (one selection, the fitted variograms stay on the device) == twx_fit_vario_points followed by
    twx_krig_points with the returned variograms, bit for bit: means, variances, variograms, bandwidths, statuses --
    including points that fail (bandwidth larger than the database) and ill-conditioned neighbourhoods (a tiny-nugget
    twin station: the second stage must route by the FITTED variogram)."""
    from topowx_amd import _lib, stationdb as sdb
    _, tmin, _ = golden_case
    stns = tmin.stns.copy()
    good = np.nonzero(np.isnan(stns[sdb.BAD]))[0]
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, sdb.StationSerialDataDb(stns, "tmin", tmin.days, None), with_obs=False)
    rng = np.random.default_rng(8)
    j = rng.choice(good.size, 40, replace=False)
    c = {k: stns[k][good] for k in (sdb.LON, sdb.LAT, sdb.ELEV, sdb.TDI)}
    lst = np.column_stack([stns[sdb.get_lst_varname(m)][good] for m in range(1, 13)])
    pt = ctx.make_pts(c[sdb.LON][j], c[sdb.LAT][j], c[sdb.ELEV][j], c[sdb.TDI][j], lst[j])
    ladder = np.array([35, 62, 91, 110, 147, 390], np.int32)               # 390 > stations - 1: those points fail
    pts = np.repeat(pt, ladder.size * 12)
    mth = np.tile(np.arange(1, 13, dtype=np.int32), j.size * ladder.size)
    nn = np.tile(np.repeat(ladder, 12), j.size)
    excl = np.repeat(j.astype(np.int32), ladder.size * 12)
    vario, used1, st1 = ctx.fit_vario_points(_lib.TMIN, pts, mth, nnghs=nn, excl=excl, rm_zero_dist=True)
    mean2, var2, _, st2, _ = ctx.krig_points(_lib.TMIN, pts, mth, nnghs=nn, vario=np.nan_to_num(vario), excl=excl, rm_zero_dist=True)
    mean, var, vfit, used, st = ctx.krigall_points(_lib.TMIN, pts, mth, nnghs=nn, excl=excl, rm_zero_dist=True)
    # the second stage alone, with the variograms krigall itself fitted: must be krigall's results bit for bit; and the
    # fitted parameters of the two SEPARATE runs too (the variogram kernel sums its bins per wave and the waves in a fixed
    # order since round 5: before, two runs agreed to rounding only)
    mean3, var3, used3, st3, _ = ctx.krig_points(_lib.TMIN, pts, mth, nnghs=nn, vario=np.nan_to_num(vfit), excl=excl, rm_zero_dist=True)
    ctx.close()
    want_st = np.where(st1 != 0, st1, st2)
    assert np.array_equal(st, want_st), (np.nonzero(st != want_st)[0][:10], st[st != want_st][:10], want_st[st != want_st][:10])
    assert (st != 0).any() and (st == 0).mean() > 0.5
    ok = st == 0
    assert np.array_equal(st3[ok], st[ok])
    assert np.array_equal(mean[ok], mean3[ok]) and np.array_equal(var[ok], var3[ok])          # bit for bit
    assert np.array_equal(used[ok], used3[ok]) and (used[~ok] == 0).all() and np.isnan(mean[~ok]).all()
    f1 = st1 == 0
    assert np.isnan(vfit[~f1]).all() and np.isfinite(vfit[f1]).all()
    assert np.array_equal(vfit[f1], vario[f1])                                                # bit for bit, run after run
    assert np.array_equal(mean[ok], mean2[ok]) and np.array_equal(var[ok], var2[ok])
