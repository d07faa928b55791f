"""GPU: the twx.interp facade classes against goldens of the reference's classes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(golden_case):
    return golden_case


def _fill(pt, grid, r, c):
    from topowx_amd import stationdb as sdb
    pt[sdb.LAT], pt[sdb.LON] = grid["lat"][r], grid["lon"][c]
    pt[sdb.ELEV], pt[sdb.TDI], pt[sdb.CLIMDIV] = grid["elev"][r, c], grid["tdi"][r, c], grid["climdiv"][r, c]
    for m in range(1, 13):
        pt["tmin%02d" % m] = grid["lst_night"][m - 1, r, c]
        pt["tmax%02d" % m] = grid["lst_day"][m - 1, r, c]


def test_station_select_krig_gwr_interp_classes(case, golden):
    from topowx_amd import stationdb as sdb
    from topowx_amd.interp import GwrTairAnom, InterpTair, KrigTair, StationSelect, build_empty_pt
    grid, tmin, _ = case
    good = np.isnan(tmin.stns[sdb.BAD])
    slct = StationSelect(tmin, good)
    ids = tmin.stns[sdb.STN_ID][good]
    # StationSelect.set_ngh_stns
    i = 0
    slct.set_ngh_stns(golden["sel_lat"][i], golden["sel_lon"][i], int(golden["sel_k"][i]), load_obs=True, obs_mth=3)
    k = int(golden["sel_k"][i])
    assert np.array_equal(slct.ngh_stns[sdb.STN_ID], ids[golden["sel_idx"][i][:k]])
    np.testing.assert_allclose(slct.ngh_wgt, golden["sel_wgt"][i][:k], rtol=1e-10)
    assert slct.ngh_obs.shape == (tmin.mth_idx[3].size, k)
    # KrigTair.krig / GwrTairAnom.gwr_mth / InterpTair.interp on the golden cells
    krig, gwr = KrigTair(slct), GwrTairAnom(slct)
    pt = build_empty_pt()
    r, c = golden["it_cell"][0]
    _fill(pt, grid, r, c)
    for m in range(1, 13):
        pt[sdb.get_lst_varname(m)] = pt["tmin%02d" % m]
    j = np.nonzero((golden["kr_cell"] == golden["it_cell"][0]).all(axis=1))[0]
    for q in j:
        mean, var = krig.krig(pt, int(golden["kr_mth"][q]))
        assert abs(mean - golden["kr_mean"][q]) < 1e-4 and abs(var - golden["kr_var"][q]) < 1e-4
        se, ci = krig.std_err_ci(mean, var)
        assert abs(se - np.sqrt(golden["kr_var"][q])) < 1e-4 and ci[0] < mean < ci[1]
    daily, norms, se = InterpTair(krig, gwr).interp(pt)
    assert np.abs(daily - golden["it_daily"][0]).max() < 1e-4 and np.abs(norms - golden["it_norms"][0]).max() < 1e-4
    assert pt[sdb.get_norm_varname(7)] == norms[6]
    d7 = gwr.gwr_mth(pt, 7)
    assert np.abs(d7 - golden["it_daily"][0][tmin.mth_idx[7]]).max() < 1e-4
    with pytest.raises(IndexError):           # stn_dists[nnghs] past the end (station_select.py:164)
        slct.set_ngh_stns(golden["sel_lat"][0], golden["sel_lon"][0], int(good.sum()))


def test_pt_interp_tair_interp_pt_and_chunk(case, golden):
    import make_golden
    from topowx_amd.interp import PtInterpTair, Tiler
    from oracle import pyoracle as orc
    grid, tmin, tmax = case
    p = PtInterpTair(tmin, make_golden.lowered_tmax(tmax))
    for i, (r, c) in enumerate(golden["lo_cell"]):
        _fill(p.a_pt, grid, r, c)
        tmin_d, tmax_d, nmin, nmax, smin, smax, ninv = p.interp_pt()
        assert ninv == golden["lo_ninv"][i] and ninv > 0
        assert np.abs(tmin_d - golden["lo_tmin"][i]).max() < 1e-4 and np.abs(tmax_d - golden["lo_tmax"][i]).max() < 1e-4
        assert np.abs(nmin - golden["lo_nmin"][i]).max() < 1e-4 and np.abs(smax - golden["lo_smax"][i]).max() < 1e-4
    # the batched work-chunk entry returns what the worker writes (step25:163-172)
    r, c = golden["lo_cell"][0]
    t = Tiler(grid, 100, 100, 5, 5)
    for k, w in t:
        if w[0, 0, 0] <= r < w[0, 0, 0] + 5 and w[1, 0, 0] <= c < w[1, 0, 0] + 5:
            break
    out = p.interp_chunk(w)
    rr, cc = int(r - w[0, 0, 0]), int(c - w[1, 0, 0])
    assert out["status"][rr, cc] == 0 and out["ninvalid"][rr, cc] == golden["lo_ninv"][0]
    dd = np.abs(out["daily_tmin"][:, rr, cc].astype(int) - orc.pack_i16(golden["lo_tmin"][0]).astype(int))
    assert dd.max() <= 1
    p.close()


def test_xval_callers(case, golden):
    from topowx_amd import stationdb as sdb
    from topowx_amd.interp import XvalTairAnom, XvalTairOverall
    grid, tmin, _ = case
    good_ids = tmin.stns[sdb.STN_ID][np.isnan(tmin.stns[sdb.BAD])]
    xo = XvalTairOverall(tmin, "tmin")
    ids = [good_ids[j] for j in golden["xv_idx"]]
    d, n, s = xo.run_interp_many(ids)
    assert np.abs(d - golden["xv_daily"]).max() < 1e-4 and np.abs(n - golden["xv_norms"]).max() < 1e-4
    d1, n1, s1 = xo.run_interp(ids[0])
    assert np.array_equal(d1, d[0])
    xo.close()
    xa = XvalTairAnom(tmin, "tmin")
    bias, mae, r2 = xa.run_xval(ids[0], np.array([35, 57, 101]))
    assert bias.shape == (3, 12) and np.all(mae > 0) and np.all((r2 >= 0) & (r2 <= 1))
    xa.close()


def test_xval_tair_anom_vs_executed_reference(case, golden_xval, orc):
    """XvalTairAnom.run_xval (optimize.py:505-545) over the full 16-bandwidth ladder against the goldens made by
    executing the reference's own run_xval, and the raw step23 call shape -- twx_gwr_points with explicit nnghs,
    the station's own index excluded and rm_zero_dist -- against executed gwr_mth series and the oracle."""
    from topowx_amd import _lib, stationdb as sdb
    from topowx_amd.interp import XvalTairAnom
    g = golden_xval
    grid, tmin, _ = case
    good = np.isnan(tmin.stns[sdb.BAD])
    good_ids = tmin.stns[sdb.STN_ID][good]
    xa = XvalTairAnom(tmin, "tmin")
    for i, j in enumerate(g["xa_stn"]):
        bias, mae, r2 = xa.run_xval(good_ids[j], g["xa_ladder"])
        assert bias.shape == (16, 12)
        assert np.abs(bias - g["xa_bias"][i]).max() < 1e-4 and np.abs(mae - g["xa_mae"][i]).max() < 1e-4
        assert np.abs(r2 - g["xa_r2"][i]).max() < 1e-6
    # raw C-ABI call in the step23 shape
    ctx = xa.ctx
    db, prm = orc.Db(tmin), orc.params()
    c = db.cols
    j, k, m = g["gx_probe"].T
    pts = ctx.make_pts(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j].T)
    pn = c["norm"][m - 1, j]
    out, used, st = ctx.gwr_points(_lib.TMIN, pts, pn, m, nnghs=k, excl=j, rm_zero_dist=True)
    assert np.all(st == 0) and np.array_equal(used, k)
    for q in range(j.size):
        nd = tmin.mth_idx[int(m[q])].size
        assert np.abs(out[q, :nd] - g["gx_series"][q, :nd]).max() < 1e-4            # executed reference
        pt = orc.make_pt(c["lon"][j[q]], c["lat"][j[q]], c["elev"][j[q]], c["tdi"][j[q]], c["lst"][:, j[q]])
        rc, want, ku, _, idx = orc.gwr_mth(db, prm, pt, pn[q], int(m[q]), nnghs=int(k[q]), excl=int(j[q]),
                                           rm_zero_dist=True)
        assert rc == 0 and ku == k[q] and j[q] not in idx
        assert np.abs(out[q, :nd] - want).max() < 1e-6                              # oracle, same arithmetic
    xa.close()


def test_variogram_fit_and_krigall(case, orc):
    """BuildKrigParams.get_krig_params / KrigTairAll.krigall / XvalTairNorm /
    StationKrigParams against the oracle's restatement of R get_vario_params (parity with gstat unpinned)."""
    from topowx_amd import stationdb as sdb
    from topowx_amd.interp import (BuildKrigParams, KrigTairAll, StationKrigParams, StationSelect, XvalTairNorm,
                                   build_nstn_bandwidths)
    grid, tmin, _ = case
    good = np.isnan(tmin.stns[sdb.BAD])
    stns = tmin.stns[good]
    db, prm = orc.Db(tmin), orc.params()
    c = db.cols
    # step22 shape: the station record is the point, it stays in its own neighbourhood
    slct = StationSelect(tmin, good)
    bkp = BuildKrigParams(slct)
    for j, m in ((3, 1), (77, 6), (150, 9), (301, 12)):
        nug, psill, rng = bkp.get_krig_params(stns[j], m)
        pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
        rc, v, _ = orc.build_krig_params(db, prm, pt, m)
        assert rc == 0
        np.testing.assert_allclose([nug, psill, rng], v, rtol=1e-6, atol=1e-9)
    skp = StationKrigParams(tmin, "tmin")
    ids = [stns[sdb.STN_ID][j] for j in (3, 77)]
    nug, psill, rng = skp.get_krig_params_many(ids)
    assert nug.shape == (2, 12) and np.all(nug > 0) and np.all(psill >= 0) and np.all(rng >= 0)
    rc, v, _ = orc.build_krig_params(db, prm, orc.make_pt(c["lon"][77], c["lat"][77], c["elev"][77], c["tdi"][77],
                                                          c["lst"][:, 77]), 6)
    np.testing.assert_allclose([nug[1, 5], psill[1, 5], rng[1, 5]], v, rtol=1e-6, atol=1e-9)
    skp.close()
    # step21 shape: leave-one-out, explicit bandwidth, fit + krige per month
    xs = StationSelect(tmin, good, rm_zero_dist_stns=True)
    ka = KrigTairAll(xs)
    for j, k in ((10, 35), (200, 57), (333, 101)):
        got = ka.krigall(stns[j], k, stns_rm=stns[sdb.STN_ID][j])
        pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
        rc, want, _ = orc.krigall(db, prm, pt, k, excl=j, rm_zero_dist=True)
        assert rc == 0 and np.abs(got - want).max() < 1e-4
    xv = XvalTairNorm(tmin, "tmin")
    ladder = build_nstn_bandwidths(35, 150, 0.10)
    err = xv.run_xval(stns[sdb.STN_ID][200], ladder)
    assert err.shape == (12, 16)
    pt = orc.make_pt(c["lon"][200], c["lat"][200], c["elev"][200], c["tdi"][200], c["lst"][:, 200])
    rc, want, _ = orc.krigall(db, prm, pt, int(ladder[5]), excl=200, rm_zero_dist=True)
    assert np.abs(err[:, 5] - (want - c["norm"][:, 200])).max() < 1e-4
    xv.close()


def test_step25_chunk_loop_equals_whole_grid(case):
    """Tiler -> wrk_chk -> PtInterpTair.interp_chunk -> tile store (the step25 worker structure)."""
    from topowx_amd import _lib, step25
    grid, tmin, tmax = case
    sub = {k: (v[:40, :60] if k in ("mask", "elev", "tdi", "climdiv") else v) for k, v in grid.items()}
    sub["lat"], sub["lon"] = grid["lat"][:40], grid["lon"][:60]
    sub["lst_night"], sub["lst_day"] = grid["lst_night"][:, :40, :60], grid["lst_day"][:, :40, :60]
    sub["mask"] = sub["mask"].copy()
    sub["mask"][:20, :20] = 0                                     # tile h00v00 has no valid cell
    stores = step25.proc_work(sub, tmin, tmax, tile_size=20, chunk_size=10, daily=True)
    assert sorted(stores) == ["h00v01", "h01v00", "h01v01", "h02v00", "h02v01"]
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    whole = ctx.interp_grid(sub, daily=True)
    ctx.close()
    for tile_id, (i, j) in (("h01v00", (0, 20)), ("h02v01", (20, 40))):
        a = stores[tile_id].a
        for k in ("norm_tmin", "se_tmax", "daily_tmin", "daily_tmax", "ninvalid", "status"):
            assert np.array_equal(a[k], whole[k][..., i:i + 20, j:j + 20]), (tile_id, k)


@pytest.mark.parametrize("out_format", ["nc4", "nc3"])
def test_step25_to_netcdf_tiles_to_monthly(case, tmp_path, out_format):
    """step25 -> netCDF tiles (8f-2, both containers) -> daily mosaic -> monthly product (8f-3), end to end: on arrays,
    and through the reference's file-level calls (step26: ``TileMosaic(fpath_mask, ...).create_dly_ann_mosaics`` /
    ``create_normals_mosaic``; step27: ``write_ds_mthly``)."""
    from topowx_amd import _lib, h5nc, ncio, step25
    from topowx_amd.interp import TairAggregate, TileMosaic, Tiler, write_ds_mthly
    if out_format == "nc4" and not h5nc.available():
        pytest.skip("libhdf5 not loadable")
    fmt = step25.NC_FORMATS[out_format]
    grid, tmin, tmax = case
    sub = {k: (v[:20, :40] if k in ("mask", "elev", "tdi", "climdiv") else v) for k, v in grid.items()}
    sub["lat"], sub["lon"] = grid["lat"][:20], grid["lon"][:40]
    sub["lst_night"], sub["lst_day"] = grid["lst_night"][:, :20, :40], grid["lst_day"][:, :20, :40]
    tdir = tmp_path / "tiles"
    tdir.mkdir()
    stores = step25.proc_work(sub, tmin, tmax, tile_size=20, chunk_size=10, daily=True, out_dir=str(tdir),
                              out_format=out_format, keep=True)
    tiles = sorted(stores)
    assert tiles == ["h00v00", "h01v00"]
    assert ncio.file_format(str(tdir / "h00v00" / "h00v00_tmin.nc")) == fmt
    back = ncio.read_tile_stores(str(tdir), tiles)
    for t in tiles:
        for k in ("daily_tmin", "daily_tmax", "norm_tmin", "se_tmax", "ninvalid"):
            np.testing.assert_array_equal(back[t].a[k], stores[t].a[k])
    info = Tiler(sub, 20, 20, 10, 10).build_tile_grid_info()
    mos = TileMosaic(info)
    dly = mos.create_dly_mosaic(tiles, "tmin", back)
    assert dly.shape == (tmin.days.size, 20, 40)
    agg = TairAggregate(tmin.days)
    m16 = agg.daily_i16_to_mthly_i16(dly)
    # numpy restatement of write_ds_mthly on the same cube
    tair = np.ma.masked_array(dly * np.float32(0.01), mask=dly == _lib.FILL_I2)
    yrs, mths = np.unique(tmin.days.YEAR), np.unique(tmin.days.MONTH)
    want = np.ma.array([np.ma.mean(tair[(tmin.days.YEAR == y) & (tmin.days.MONTH == m)], axis=0, dtype=float)
                        for y in yrs for m in mths])
    want = np.around(np.ma.getdata(np.ma.round(want, 2)) / np.float32(0.01))
    want = np.where(np.ma.getmaskarray(tair).all(axis=0)[None], _lib.FILL_I2, want).astype(np.int16)
    np.testing.assert_array_equal(m16, want)
    agg.close()

    # ---- the same through files, with the reference's call shapes --------------------------------------------------
    fmask = str(tmp_path / "mask.nc")
    ds = ncio.open_dataset(fmask, "w", fmt)
    ds.createDimension("lat", 20)
    ds.createDimension("lon", 40)
    ds.createVariable("lat", "f8", ("lat",))[:] = sub["lat"]
    ds.createVariable("lon", "f8", ("lon",))[:] = sub["lon"]
    ds.createVariable("mask", "i1", ("lat", "lon"))[:] = sub["mask"].astype(np.int8)
    ds.close()
    fm = TileMosaic(fmask, 20, 20, 10, 10)                                     # step26:30-31
    y0, y1 = int(yrs[0]), int(yrs[-1])
    paths = fm.create_dly_ann_mosaics(tiles, "tmin", str(tdir), str(tmp_path / "daily"), y0, y1, "9.9.9", 50000000,
                                      format=fmt)                             # step26:52-54
    assert [os.path.basename(p) for p in paths] == ["tmin_%d.nc" % y for y in yrs]
    nyr = yrs.size
    for q, (p, y) in enumerate(zip(paths, yrs)):
        ds_dly = ncio.open_dataset(p)
        rows = np.nonzero(tmin.days.YEAR == y)[0]
        np.testing.assert_array_equal(ds_dly.variables["tmin"][:], dly[rows])
        assert ds_dly.variables["time"].units == "days since 1948-1-1 0:0:0" and "9.9.9" in ds_dly.history
        fo = str(tmp_path / ("tmin_mthly_%d.nc" % y))
        write_ds_mthly(ds_dly, fo, "tmin", int(y), "9.9.9", format=fmt)        # step27:33-36
        ds_dly.close()
        ds_m = ncio.open_dataset(fo)
        np.testing.assert_array_equal(ds_m.variables["tmin"][:], want[12 * q:12 * q + 12])
        assert ds_m.variables["tmin"].shape == (12, 20, 40) and ds_m.title.endswith(str(y))
        if fmt == "NETCDF4":
            assert ds_m.variables["tmin"].filters()["zlib"] and ds_m.variables["tmin"].chunking() == [1, 20, 40]
        ds_m.close()
    assert nyr >= 1
    fn = fm.create_normals_mosaic(tiles, "tmax", str(tdir), str(tmp_path / "normals_tmax.nc"), "9.9.9", format=fmt)
    pn, ps = mos.create_normals_mosaic(tiles, "tmax", back)
    ds_n = ncio.open_dataset(fn)
    np.testing.assert_array_equal(ds_n.variables["tmax_normal"][:], pn)
    np.testing.assert_array_equal(ds_n.variables["tmax_se"][:], ps)
    ds_n.close()
