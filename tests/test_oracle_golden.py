"""CPU oracle vs golden vectors produced by executing the reference's own source
slices (tests/golden/make_golden.py).  Pins SURVEY.md rows a1-a4, a7-a9, a12."""
import numpy as np
import pytest

from topowx_amd import stationdb as sdb


def _pt(orc, grid, r, c, var):
    lst = grid["lst_night" if var == "tmin" else "lst_day"][:, r, c]
    return orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], lst)


@pytest.fixture(scope="module")
def dbs(orc, golden_case):
    grid, tmin, tmax = golden_case
    return grid, orc.Db(tmin), orc.Db(tmax)


def test_haversine(orc, golden):
    d = orc.grt_circle_dist(golden["hv_lon1"], golden["hv_lat1"], golden["hv_lon2"], golden["hv_lat2"])
    np.testing.assert_allclose(d, golden["hv_dist"], rtol=1e-14, atol=1e-12)
    assert np.all(d[:3] == 0.0)


def test_station_select(orc, golden, dbs):
    _, dbn, _ = dbs
    for i in range(golden["sel_k"].size):
        k = int(golden["sel_k"][i])
        rc, idx, dist, wgt = orc.select(dbn, golden["sel_lat"][i], golden["sel_lon"][i], k,
                                        int(golden["sel_excl"][i]), bool(golden["sel_rmz"][i]))
        assert rc == 0
        assert np.array_equal(idx, golden["sel_idx"][i][:k])          # bit-exact indices
        np.testing.assert_allclose(dist, golden["sel_dist"][i][:k], rtol=1e-13, atol=1e-11)
        np.testing.assert_allclose(wgt, golden["sel_wgt"][i][:k], rtol=1e-11, atol=1e-13)


def test_nnghs_vario_krig(orc, golden, dbs):
    grid, dbn, _ = dbs
    prm = orc.params()
    for i in range(golden["kr_mth"].size):
        r, c = golden["kr_cell"][i]
        m = int(golden["kr_mth"][i])
        pt = _pt(orc, grid, r, c, "tmin")
        rc, mean, var, used, _ = orc.krig(dbn, prm, pt, m)
        assert rc == 0
        assert used == golden["kr_nnghs"][i]                            # a3 exact
        # orchestration from the reference slice + independent numpy UK (augmented system)
        assert abs(mean - golden["kr_mean"][i]) < 1e-7
        assert abs(var - golden["kr_var"][i]) < 1e-7
        rc, _, ka, _, _ = orc.gwr_mth(dbn, prm, pt, 0.0, m)
        assert rc == 0 and ka == golden["kr_nnghs_anom"][i]


def test_vario_smoothing(orc, golden, dbs):
    import ctypes as C
    grid, dbn, _ = dbs
    for i in range(golden["kr_mth"].size):
        r, c = golden["kr_cell"][i]
        m = int(golden["kr_mth"][i])
        k = int(golden["kr_nnghs"][i])
        rc, idx, _, wgt = orc.select(dbn, grid["lat"][r], grid["lon"][c], k)
        v = np.zeros(3)
        dp = C.POINTER(C.c_double)
        rc = orc.lib().orc_smooth_vario(
            dbn.cols["vario_nug"][m - 1].ctypes.data_as(dp), dbn.cols["vario_psill"][m - 1].ctypes.data_as(dp),
            dbn.cols["vario_rng"][m - 1].ctypes.data_as(dp), idx.ctypes.data_as(C.POINTER(C.c_int32)),
            wgt.ctypes.data_as(dp), C.c_int(k), v.ctypes.data_as(dp))
        assert rc == 0
        np.testing.assert_allclose(v, golden["kr_vario"][i], rtol=1e-12)


def test_krig_explicit_args(orc, golden, dbs):
    grid, dbn, _ = dbs
    prm = orc.params()
    r, c = golden["krx_cell"]
    pt = _pt(orc, grid, r, c, "tmin")
    got = [orc.krig(dbn, prm, pt, 3, nnghs=57)[1:3],
           orc.krig(dbn, prm, pt, 3, nnghs=40, vario=(0.2, 1.1, 35.0))[1:3],
           orc.krig(dbn, prm, pt, 3, vario=(0.3, 0.9, 0.0))[1:3],      # range 0 -> pure nugget
           orc.krig(dbn, prm, pt, 3, excl=int(golden["krx_rm"]))[1:3]]
    np.testing.assert_allclose(np.array(got), golden["krx"], atol=1e-7, rtol=0)


def test_gwr_series(orc, golden):
    rc, z = orc.gwr_hat(golden["gs_X"], golden["gs_w"], golden["gs_x"])
    assert rc == 0
    got = golden["gs_y"] @ z
    # the reference inverts the raw, badly scaled 6x6 (np.linalg.inv); agreement is
    # limited by ITS conditioning, and is far inside the 1e-4 degC parity bar
    np.testing.assert_allclose(got, golden["gs_out"], atol=1e-8, rtol=0)


def test_interp_one_variable(orc, golden, dbs):
    grid, dbn, _ = dbs
    prm = orc.params()
    for i, (r, c) in enumerate(golden["it_cell"]):
        rc, daily, norms, se = orc.interp(dbn, prm, _pt(orc, grid, r, c, "tmin"))
        assert rc == 0
        np.testing.assert_allclose(norms, golden["it_norms"][i], atol=1e-7, rtol=0)
        np.testing.assert_allclose(se, golden["it_se"][i], atol=1e-7, rtol=0)
        np.testing.assert_allclose(daily, golden["it_daily"][i], atol=5e-6, rtol=0)


def test_interp_leave_one_out(orc, golden, dbs):
    _, dbn, _ = dbs
    prm = orc.params()
    c = dbn.cols
    for i, j in enumerate(golden["xv_idx"]):
        pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
        rc, daily, norms, se = orc.interp(dbn, prm, pt, excl=int(j), rm_zero_dist=True)
        assert rc == 0
        np.testing.assert_allclose(norms, golden["xv_norms"][i], atol=1e-7, rtol=0)
        np.testing.assert_allclose(se, golden["xv_se"][i], atol=1e-7, rtol=0)
        np.testing.assert_allclose(daily, golden["xv_daily"][i], atol=5e-6, rtol=0)


def test_fixer(orc, golden):
    for i in range(golden["fx_n"].size):
        rc, tmin, tmax, n = orc.fixer(golden["fx_in_min"][i], golden["fx_in_max"][i])
        assert rc == 0 and n == golden["fx_n"][i]
        np.testing.assert_allclose(tmin, golden["fx_min"][i], rtol=0, atol=1e-13)
        np.testing.assert_allclose(tmax, golden["fx_max"][i], rtol=0, atol=1e-13)
    # window with no valid day -> the reference raises (interp_tair.py:192)
    a = np.zeros(10)
    rc, _, _, _ = orc.fixer(a, a - 1.0)
    assert rc == 5


def test_pack(orc, golden):
    assert np.array_equal(orc.pack_i16(golden["pk_in"]), golden["pk_out"])


def test_pack_identity(orc):
    """The GPU packs a value as (int16) rint(100 x) (twx_daily.h: pack_i16); the reference's expression
    np.round(x, 2) / np.float32(0.01) truncated on assignment (step25:163-164) -- which the oracle keeps literally --
    is the same integer for EVERY n = rint(100 x) an int16 product can hold (and twice that range): exhaustive."""
    n = np.arange(-70000, 70001).astype(np.float64)
    c = np.float64(np.float32(0.01))
    assert np.array_equal(np.trunc((n / 100.0) / c), n)
    # through the oracle's own packing, on values straddling every rounding boundary of the int16 range
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-327.6, 327.6, 200000), (np.arange(-32760, 32760) + 0.5) / 100.0,
                        np.nextafter((np.arange(-32760, 32760) + 0.5) / 100.0, 1e9), golden_like_values()])
    assert np.array_equal(orc.pack_i16(x), np.rint(x * 100.0).astype(np.int64).astype(np.int16))


def golden_like_values():
    return np.array([12.345, -12.345, -0.015, 25.675, 0.0, -0.0, 0.005, -0.005, 327.67, -327.67, 1e-300])


def test_ladder(golden):
    from topowx_amd.synth import NNGH_LADDER
    assert np.array_equal(golden["ladder"], NNGH_LADDER)


@pytest.mark.parametrize("pre", ["pt", "lo"])
def test_interp_pt_both_variables(orc, golden, golden_case, dbs, pre):
    """PtInterpTair.interp_pt incl. fixer + normals recompute, through the grid entry.
    'lo' = Tmax DB shifted down so the fixer and the normals recompute really run."""
    grid, dbn, dbx = dbs
    if pre == "lo":
        import make_golden
        dbx = orc.Db(make_golden.lowered_tmax(golden_case[2]))
        assert golden["lo_ninv"].min() > 0
    golden = {k.replace(pre + "_", "pt_", 1): golden[k] for k in golden.files if k.startswith(pre + "_")}
    prm = orc.params()
    for i, (r, c) in enumerate(golden["pt_cell"]):
        out = orc.interp_grid(dbn, dbx, prm, grid, daily=True, rows=slice(r, r + 1), cols=slice(c, c + 1))
        assert out["status"][0, 0] == 0
        assert out["ninvalid"][0, 0] == golden["pt_ninv"][i]
        np.testing.assert_allclose(out["norm_tmin"][:, 0, 0], golden["pt_nmin"][i].astype(np.float32), rtol=2e-6)
        np.testing.assert_allclose(out["norm_tmax"][:, 0, 0], golden["pt_nmax"][i].astype(np.float32), rtol=2e-6)
        np.testing.assert_allclose(out["se_tmin"][:, 0, 0], golden["pt_smin"][i].astype(np.float32), rtol=2e-6)
        for v, key in (("tmin", "pt_tmin"), ("tmax", "pt_tmax")):
            want = orc.pack_i16(golden[key][i])
            got = out["daily_" + v][:, 0, 0]
            # 1e-6 degC differences can flip one LSB at a 0.005 rounding boundary
            assert np.max(np.abs(got.astype(int) - want.astype(int))) <= 1
            assert np.mean(got == want) > 0.999


def test_xval_anom_oracle_vs_executed_run_xval(orc, golden_xval, golden_case):
    """a13: the step23 call shape -- gwr_mth(stn, mth, nnghs, stns_rm = own id) on a
    StationSelect(rm_zero_dist_stns=True) -- and the bias / MAE / r^2 of XvalTairAnom.run_xval
    (optimize.py:505-545), against goldens made by executing the reference's own code."""
    g = golden_xval
    _, tmin, _ = golden_case
    db, prm = orc.Db(tmin), orc.params()
    c = db.cols
    for q, (j, k, m) in enumerate(g["gx_probe"]):
        pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
        rc, out, ku, _, idx = orc.gwr_mth(db, prm, pt, c["norm"][m - 1, j], int(m), nnghs=int(k), excl=int(j),
                                          rm_zero_dist=True)
        assert rc == 0 and ku == k and j not in idx
        np.testing.assert_allclose(out, g["gx_series"][q, :out.size], rtol=0, atol=1e-9)
    # full ladder for one station: the statistics as the reference computes them
    j = int(g["xa_stn"][1])
    pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
    for x, k in enumerate(g["xa_ladder"]):
        for m in range(1, 13):
            nrm = c["norm"][m - 1, j]
            rc, out, _, _, _ = orc.gwr_mth(db, prm, pt, nrm, m, nnghs=int(k), excl=j, rm_zero_dist=True)
            assert rc == 0
            xval_anom = db.obs[db.day_month == m, j].astype(np.float64) - nrm
            difs = (out - nrm) - xval_anom
            assert abs(difs.mean() - g["xa_bias"][1][x, m - 1]) < 1e-6
            assert abs(np.abs(difs).mean() - g["xa_mae"][1][x, m - 1]) < 1e-6
            assert abs(np.corrcoef(out - nrm, xval_anom)[0, 1] ** 2 - g["xa_r2"][1][x, m - 1]) < 1e-9


def test_oracle_stns_rm_arrays_match_the_executed_reference(orc, golden_case):
    """``stns_rm`` as an array of ids (station_select.py:74-103): the oracle with an exclusion LIST against
    tests/golden/golden_rm_v1.npz (made by executing the reference's StationSelect / KrigTair / GwrTairAnom with id arrays)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_rm_v1.npz"))
    grid, tmin, _ = golden_case
    db, prm = orc.Db(tmin), orc.params()
    for q in range(g["k"].size):
        k = int(g["k"][q])
        with orc.exclusions(g["excl"][q]):
            rc, idx, dist, wgt = orc.select(db, g["lat"][q], g["lon"][q], k, rm_zero_dist=bool(g["rmz"][q]))
        assert rc == 0 and np.array_equal(idx, g["idx"][q][:k])                               # bit-exact
        np.testing.assert_allclose(dist, g["dist"][q][:k], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(wgt, g["wgt"][q][:k], rtol=1e-10, atol=1e-12)
    for i, (r, c) in enumerate(g["kr_cell"]):
        pt = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c])
        mth = int(g["kr_mth"][i])
        with orc.exclusions(g["kr_excl"][i]):
            rc, mean, var, used, _ = orc.krig(db, prm, pt, mth)
            rc2, series, ka, _, _ = orc.gwr_mth(db, prm, pt, float(g["kr_mean"][i]), mth)
        assert rc == 0 and rc2 == 0
        assert abs(mean - g["kr_mean"][i]) < 1e-6 and abs(var - g["kr_var"][i]) < 1e-6
        n = int(g["gw_len"][i])
        assert np.abs(series[:n] - g["gw_series"][i, :n]).max() < 1e-8
    # and the setting does not outlive its block
    rc, idx, _, _ = orc.select(db, g["lat"][0], g["lon"][0], 35)
    assert np.intersect1d(idx, g["excl"][0][g["excl"][0] >= 0]).size > 0
