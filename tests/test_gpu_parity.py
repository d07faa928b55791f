"""GPU parity: libtwxhip (through the C ABI) vs the CPU oracle and the golden
vectors.  Tolerance from BASELINE.json: 1e-4 degC on temperatures, integer
indices / bandwidths / ninvalid bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4  # degC (north_star)


@pytest.fixture(scope="module")
def env(orc, golden, golden_case):
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    yield dict(lib=_lib, ctx=ctx, grid=grid, tmin=tmin, tmax=tmax, dbn=orc.Db(tmin), dbx=orc.Db(tmax),
               prm=orc.params())
    ctx.close()


def _pts(ctx, grid, cells, var):
    r, c = cells[:, 0], cells[:, 1]
    lst = grid["lst_night" if var == "tmin" else "lst_day"][:, r, c].T
    return ctx.make_pts(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], lst)


def test_knn_matches_reference_station_select(env, golden):
    ctx, lib = env["ctx"], env["lib"]
    for k in (35, 100, 147):
        for rmz in (0, 1):
            sel = np.nonzero((golden["sel_k"] == k) & (golden["sel_rmz"] == rmz))[0]
            idx, dist, wgt, st = ctx.knn(lib.TMIN, golden["sel_lon"][sel], golden["sel_lat"][sel], k,
                                         excl=golden["sel_excl"][sel], rm_zero_dist=bool(rmz))
            assert np.all(st == 0)
            assert np.array_equal(idx, golden["sel_idx"][sel][:, :k])         # bit-exact
            np.testing.assert_allclose(dist, golden["sel_dist"][sel][:, :k], rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(wgt, golden["sel_wgt"][sel][:, :k], rtol=1e-10, atol=1e-12)


def test_knn_ties_rank_in_table_order(golden_case, orc):
    """Equal distances (co-located stations: step20 removes exact duplicates of ID, not of place) rank in table order,
    as the oracle's selection does, wherever the tie falls -- inside the neighbourhood, on its last rank, across it.
    k_select ranks by counting the strictly nearer candidates and ranks again, ties in list order, only when that
    leaves a hole; every k from 3 to 60 moves the boundary across the triplets of this table."""
    from topowx_amd import _lib, stationdb as sdb
    grid, tmin, _ = golden_case
    stns = tmin.stns.copy()
    rng = np.random.default_rng(4)
    src = rng.choice(stns.size, 60, replace=False)
    for a in src[:40]:                                        # 40 twins and 20 triplets, scattered over the table
        b = (a + 1 + rng.integers(0, stns.size - 1)) % stns.size
        stns[sdb.LON][b], stns[sdb.LAT][b] = stns[sdb.LON][a], stns[sdb.LAT][a]
    for a in src[40:]:
        for _ in range(2):
            b = (a + 1 + rng.integers(0, stns.size - 1)) % stns.size
            stns[sdb.LON][b], stns[sdb.LAT][b] = stns[sdb.LON][a], stns[sdb.LAT][a]
    db = sdb.StationDataWrkChk(stns, "tmin", tmin.days, None)
    odb = orc.Db(db)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, db, with_obs=False)
    cells = np.argwhere(np.asarray(grid["mask"]) != 0)[::151][:24]
    lon, lat = np.asarray(grid["lon"])[cells[:, 1]], np.asarray(grid["lat"])[cells[:, 0]]
    ties = 0
    for k in range(3, 61):
        idx, dist, _, st = ctx.knn(_lib.TMIN, lon, lat, k)
        assert np.all(st == 0)
        for i in range(len(cells)):
            rc, want, _, _ = orc.select(odb, lat[i], lon[i], k)
            assert rc == 0 and np.array_equal(idx[i], want), (k, i, idx[i], want)
            ties += int(np.any(np.diff(dist[i]) == 0.0))
    ctx.close()
    assert ties >= 30, ties


def test_krig_points_golden(env, golden):
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    pts = _pts(ctx, grid, golden["kr_cell"], "tmin")
    mean, var, used, st, ngh = ctx.krig_points(lib.TMIN, pts, golden["kr_mth"], want_idx=True)
    assert np.all(st == 0)
    assert np.array_equal(used, golden["kr_nnghs"])
    assert np.abs(mean - golden["kr_mean"]).max() < TOL
    assert np.abs(var - golden["kr_var"]).max() < TOL
    # neighbours used == oracle's (ascending station index)
    for i in range(pts.size):
        r, c = golden["kr_cell"][i]
        rc, idx, _, _ = __import__("oracle.pyoracle", fromlist=["x"]).select(env["dbn"], grid["lat"][r], grid["lon"][c], int(used[i]))
        assert np.array_equal(ngh[i, :used[i]], idx)
        assert np.all(ngh[i, used[i]:] == -1)


def test_krig_points_explicit_arguments(env, golden, orc):
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    pts = _pts(ctx, grid, np.tile(golden["krx_cell"], (4, 1)), "tmin")
    nan = np.nan
    mean, var, used, st, _ = ctx.krig_points(
        lib.TMIN, pts, 3, nnghs=[57, 40, 0, 0],
        vario=[[nan, nan, nan], [0.2, 1.1, 35.0], [0.3, 0.9, 0.0], [nan, nan, nan]],
        excl=[-1, -1, -1, int(golden["krx_rm"])])
    assert np.all(st == 0)
    np.testing.assert_allclose(np.column_stack([mean, var]), golden["krx"], atol=TOL, rtol=0)
    assert used[0] == 57 and used[1] == 40


def test_grid_normals_vs_oracle(env, orc):
    ctx, grid = env["ctx"], env["grid"]
    rs, cs = slice(3, 43), slice(50, 90)      # 40x40 cells, not aligned to the 8-cell tiles
    want = orc.interp_grid(env["dbn"], env["dbx"], env["prm"], grid, daily=False, nthreads=8, rows=rs, cols=cs)
    got = ctx.interp_grid(grid, daily=False, rows=rs, cols=cs)
    assert np.array_equal(got["status"], want["status"]) and np.all(got["status"] == 0)
    for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
        assert np.abs(got[k].astype(np.float64) - want[k]).max() < TOL, k
    assert np.all(got["ninvalid"] == 0)


def test_grid_mask_and_fill_values(env, orc):
    ctx, lib = env["ctx"], env["lib"]
    grid = dict(env["grid"])
    mask = np.zeros_like(grid["mask"])
    mask[5:9, 5:30:3] = 1
    grid["mask"] = mask
    rs, cs = slice(0, 16), slice(0, 40)
    got = ctx.interp_grid(grid, variables=("tmin",), rows=rs, cols=cs)
    m = mask[rs, cs] != 0
    assert np.all(got["status"][~m] == -1) and np.all(got["status"][m] == 0)
    assert np.all(got["norm_tmin"][:, ~m] == lib.FILL_F4) and np.all(got["se_tmin"][:, ~m] == lib.FILL_F4)
    assert np.all(got["ninvalid"][~m] == lib.FILL_I4)
    want = orc.interp_grid(env["dbn"], None, env["prm"], grid, rows=rs, cols=cs)
    assert np.abs(got["norm_tmin"][:, m].astype(np.float64) - want["norm_tmin"][:, m]).max() < TOL


def test_gwr_points_golden(env, golden, orc):
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    cells = golden["it_cell"]
    pts = _pts(ctx, grid, cells, "tmin")
    for m in (1, 2, 7):
        pn = golden["it_norms"][:, m - 1]
        out, used, st = ctx.gwr_points(lib.TMIN, pts, pn, m)
        assert np.all(st == 0)
        idx = env["tmin"].mth_idx[m]
        np.testing.assert_allclose(out[:, :idx.size], golden["it_daily"][:, idx], atol=TOL, rtol=0)
        for i in range(pts.size):
            rc, _, ka, _, _ = orc.gwr_mth(env["dbn"], env["prm"], orc.make_pt(pts["lon"][i], pts["lat"][i], pts["elev"][i],
                                                                          pts["tdi"][i], pts["lst"][i]), 0.0, m)
            assert used[i] == ka


def test_interp_points_golden_and_xval(env, golden):
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    d, norms, se, st = ctx.interp_points(lib.TMIN, _pts(ctx, grid, golden["it_cell"], "tmin"))
    assert np.all(st == 0)
    assert np.abs(norms - golden["it_norms"]).max() < TOL and np.abs(se - golden["it_se"]).max() < TOL
    assert np.abs(d - golden["it_daily"]).max() < TOL
    # leave-one-out (XvalTairOverall.run_interp, optimize.py:579-604)
    c, j = env["dbn"].cols, golden["xv_idx"]
    pts = ctx.make_pts(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j].T)
    d, norms, se, st = ctx.interp_points(lib.TMIN, pts, excl=j, rm_zero_dist=True)
    assert np.all(st == 0)
    assert np.abs(norms - golden["xv_norms"]).max() < TOL and np.abs(d - golden["xv_daily"]).max() < TOL


@pytest.mark.parametrize("lowered", [False, True])
def test_grid_daily_with_fixer_vs_oracle(env, orc, golden_case, lowered):
    import make_golden
    lib, grid = env["lib"], env["grid"]
    tmax = make_golden.lowered_tmax(golden_case[2]) if lowered else golden_case[2]
    ctx = lib.Context()
    ctx.set_stations(lib.TMIN, golden_case[1])
    ctx.set_stations(lib.TMAX, tmax)
    dbx = orc.Db(tmax)
    rs, cs = slice(60, 70), slice(11, 24)
    want = orc.interp_grid(env["dbn"], dbx, env["prm"], grid, daily=True, nthreads=8, rows=rs, cols=cs)
    got = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
    ctx.close()
    assert np.array_equal(got["status"], want["status"]) and np.all(got["status"] == 0)
    assert np.array_equal(got["ninvalid"], want["ninvalid"])            # exact
    assert (want["ninvalid"].max() > 0) == lowered
    for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
        assert np.abs(got[k].astype(np.float64) - want[k]).max() < TOL, k
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(got[k].astype(int) - want[k].astype(int))
        assert dd.max() <= 1 and (dd == 0).mean() > 0.9999          # +-1 LSB at a rounding boundary at most


def test_interp_pt_goldens_through_grid(env, golden, golden_case):
    """PtInterpTair.interp_pt goldens (reference orchestration incl. fixer)."""
    import make_golden
    lib, grid = env["lib"], env["grid"]
    ctx = lib.Context()
    ctx.set_stations(lib.TMIN, golden_case[1])
    ctx.set_stations(lib.TMAX, make_golden.lowered_tmax(golden_case[2]))
    from oracle import pyoracle
    for i, (r, c) in enumerate(golden["lo_cell"]):
        got = ctx.interp_grid(grid, daily=True, rows=slice(r, r + 1), cols=slice(c, c + 1))
        assert got["status"][0, 0] == 0 and got["ninvalid"][0, 0] == golden["lo_ninv"][i]
        assert np.abs(got["norm_tmin"][:, 0, 0] - golden["lo_nmin"][i]).max() < TOL
        assert np.abs(got["norm_tmax"][:, 0, 0] - golden["lo_nmax"][i]).max() < TOL
        for v, key in (("tmin", "lo_tmin"), ("tmax", "lo_tmax")):
            dd = np.abs(got["daily_" + v][:, 0, 0].astype(int) - pyoracle.pack_i16(golden[key][i]).astype(int))
            assert dd.max() <= 1 and (dd == 0).mean() > 0.999
    ctx.close()


def test_fix_pair_and_pack(env, golden, orc):
    ctx = env["ctx"]
    nd = ctx.ndays
    a = np.tile(golden["fx_in_min"], (1, 3))[:, :nd]
    b = np.tile(golden["fx_in_max"], (1, 3))[:, :nd]
    fa, fb, ninv, nmin, nmax, st = ctx.fix_pair(a, b)
    for i in range(a.shape[0]):
        rc, oa, ob, on = orc.fixer(a[i], b[i])
        assert st[i] == rc == 0 and ninv[i] == on
        np.testing.assert_allclose(fa[i], oa, atol=1e-12, rtol=0)
        np.testing.assert_allclose(fb[i], ob, atol=1e-12, rtol=0)
        if on > 0:
            np.testing.assert_allclose(nmin[i], orc.recompute_norms(oa, env["dbn"].day_month, env["dbn"].day_year), atol=1e-10)
    # window without a valid day: the reference raises (interp_tair.py:192)
    z = np.zeros((1, nd))
    _, _, _, _, _, st = ctx.fix_pair(z, z - 1.0)
    assert st[0] == 5
    assert np.array_equal(ctx.pack_i16(golden["pk_in"]), golden["pk_out"])     # bit-exact


def test_failure_statuses(env, orc):
    """Cells the reference would abandon keep their fill values (step25:154-160)."""
    from topowx_amd import stationdb as sdb, synth
    lib, grid = env["lib"], env["grid"]
    # (1) fewer than init_nnghs+1 stations -> IndexError in the reference
    few = synth.make_stations(grid["bbox"], 60, 5, "tmin")
    ctx = lib.Context()
    ctx.set_stations(lib.TMIN, few, with_obs=False)
    got = ctx.interp_grid(grid, variables=("tmin",), rows=slice(0, 4), cols=slice(0, 4))
    want = orc.interp_grid(orc.Db(few), None, env["prm"], grid, rows=slice(0, 4), cols=slice(0, 4))
    assert np.all(got["status"] == 1) and np.array_equal(got["status"], want["status"])
    assert np.all(got["norm_tmin"] == lib.FILL_F4)
    # (2) no finite optim_nnghs among the neighbours
    st = synth.make_stations(grid["bbox"], 300, 6, "tmin")
    st.stns[sdb.get_optim_varname(5)][:] = np.nan
    ctx.set_stations(lib.TMIN, st, with_obs=False)
    got = ctx.interp_grid(grid, variables=("tmin",), rows=slice(0, 2), cols=slice(0, 2))
    want = orc.interp_grid(orc.Db(st), None, env["prm"], grid, rows=slice(0, 2), cols=slice(0, 2))
    assert np.all(got["status"] == 2) and np.array_equal(got["status"], want["status"])
    # (3) no finite variogram parameters
    st = synth.make_stations(grid["bbox"], 300, 6, "tmin")
    st.stns[sdb.get_krigparam_varname(2, sdb.VARIO_NUG)][:] = np.nan
    ctx.set_stations(lib.TMIN, st, with_obs=False)
    got = ctx.interp_grid(grid, variables=("tmin",), rows=slice(0, 2), cols=slice(0, 2))
    assert np.all(got["status"] == 3)
    # (4) duplicate station location, zero nugget -> singular kriging matrix
    st = synth.make_stations(grid["bbox"], 300, 6, "tmin")
    j = np.argmin((st.stns[sdb.LON] - grid["lon"][0]) ** 2 + (st.stns[sdb.LAT] - grid["lat"][0]) ** 2)
    i = (j + 1) % st.stns.size
    st.stns[sdb.LON][i], st.stns[sdb.LAT][i] = st.stns[sdb.LON][j], st.stns[sdb.LAT][j]
    pts = ctx.make_pts(grid["lon"][0], grid["lat"][0], 1500.0, 30.0, np.zeros(12))
    ctx.set_stations(lib.TMIN, st, with_obs=False)
    _, _, _, s4, _ = ctx.krig_points(lib.TMIN, pts, 1, nnghs=40, vario=[0.0, 1.0, 40.0])
    assert s4[0] == 4
    ctx.close()


def test_cell_on_station_is_exact_interpolator(env):
    ctx, lib = env["ctx"], env["lib"]
    c = env["dbn"].cols
    j = 17
    pts = ctx.make_pts(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
    mean, var, _, st, _ = ctx.krig_points(lib.TMIN, pts, 6)
    # pair distances are kept in fp32 registers on the GPU: exactness holds to ~1e-7 degC
    assert st[0] == 0 and abs(mean[0] - c["norm"][5, j]) < 1e-6 and abs(var[0]) < 1e-6


def test_device_pointer_entry_matches_host_entry(env):
    """twx_interp_grid_dev (inputs resident in HBM, caller's stream) == twx_interp_grid."""
    import torch
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    rs, cs = slice(8, 40), slice(16, 64)
    host = ctx.interp_grid(grid, variables=("tmin",), rows=rs, cols=cs)
    a = ctx.grid_arrays(grid, rs, cs)
    Y, X = a["mask"].shape
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(v).to(dev) for k, v in a.items()}
    norm = torch.full((12, Y, X), float(lib.FILL_F4), dtype=torch.float32, device=dev)
    se = torch.full((12, Y, X), float(lib.FILL_F4), dtype=torch.float32, device=dev)
    stat = torch.full((Y, X), -1, dtype=torch.int32, device=dev)
    g = lib.TwxGrid(Y, X, d["mask"].data_ptr(), d["lat"].data_ptr(), d["lon"].data_ptr(), d["elev"].data_ptr(),
                    d["tdi"].data_ptr(), d["climdiv"].data_ptr(), d["lst_night"].data_ptr(), None)
    o = lib.TwxGridOut(norm.data_ptr(), se.data_ptr(), None, None, None, None, None, stat.data_ptr())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.interp_grid_dev(g, o, lib.VAR_TMIN_BIT, s.cuda_stream)
    s.synchronize()
    assert np.array_equal(norm.cpu().numpy(), host["norm_tmin"]) and np.array_equal(se.cpu().numpy(), host["se_tmin"])
    assert np.array_equal(stat.cpu().numpy(), host["status"])
    t = ctx.timing()
    assert t["cells"] == Y * X and t["uk_solves"] == Y * X * 12 and t["uk_ms"] > 0 and t["total_ms"] >= t["uk_ms"]
    k = ctx.last_bandwidths(lib.TMIN)
    assert k.shape == (Y * X, 12) and k.min() >= 35 and k.max() <= 147


def test_all_masked_and_nan_predictors(env):
    """A chunk without any valid cell is a no-op; a NaN predictor fails only its own cell
    (np.seterr(all='raise') -> FloatingPointError -> fill values, step25:154-160,319)."""
    ctx, lib = env["ctx"], env["lib"]
    grid = dict(env["grid"])
    grid["mask"] = np.zeros_like(grid["mask"])
    got = ctx.interp_grid(grid, rows=slice(0, 20), cols=slice(0, 30))
    assert np.all(got["status"] == -1) and np.all(got["norm_tmin"] == lib.FILL_F4)
    grid = dict(env["grid"])
    grid["elev"] = grid["elev"].copy()
    grid["lst_day"] = grid["lst_day"].copy()
    grid["elev"][3, 4] = np.nan
    grid["lst_day"][6, 5, 9] = np.nan
    got = ctx.interp_grid(grid, rows=slice(0, 12), cols=slice(0, 12))
    bad = np.zeros((12, 12), bool)
    bad[3, 4] = bad[5, 9] = True
    assert np.all(got["status"][bad] == 4) and np.all(got["status"][~bad] == 0)
    assert np.all(got["norm_tmin"][:, bad] == lib.FILL_F4) and np.all(got["norm_tmax"][:, bad] == lib.FILL_F4)
    assert np.all(np.isfinite(got["norm_tmin"][:, ~bad])) and np.all(got["ninvalid"][bad] == lib.FILL_I4)


def test_failures_of_two_kinds_keep_the_reference_order(env, orc, golden_case):
    """A cell with a NaN predictor in a database too small for some months' bandwidths fails twice: its kriging with a
    floating-point error in month 1, its station selection in a later month.  The reference walks the months in order
    (interp_tair.py:429-437), so the FIRST failure is the one it reports -- the kriging's (found by tests/tools/gpu_soak.py,
    seed 5043: the selection kernel, which runs first on the GPU, used to report its own)."""
    from topowx_amd import stationdb as sdb
    lib = env["lib"]
    grid = dict(env["grid"])
    for name in ("elev", "tdi", "lst_night", "lst_day"):
        grid[name] = grid[name].copy()
    grid["elev"][2:4, 5:7] = np.nan
    grid["tdi"][8, 1] = np.nan
    grid["lst_night"][7, 10, 3] = np.nan                                 # August only
    rs, cs = slice(0, 12), slice(0, 12)
    for nkeep in (104, 125, 150):                                        # too few stations for all / some / the largest bandwidths
        dbn = sdb.StationDataWrkChk(golden_case[1].stns[:nkeep].copy(), "tmin", golden_case[1].days, golden_case[1].var[:, :nkeep].copy())
        dbx = sdb.StationDataWrkChk(golden_case[2].stns[:nkeep].copy(), "tmax", golden_case[2].days, golden_case[2].var[:, :nkeep].copy())
        ctx = lib.Context()
        ctx.set_stations(lib.TMIN, dbn)
        ctx.set_stations(lib.TMAX, dbx)
        for daily in (False, True):
            got = ctx.interp_grid(grid, daily=daily, rows=rs, cols=cs)
            want = orc.interp_grid(orc.Db(dbn), orc.Db(dbx), env["prm"], grid, daily=daily, nthreads=8, rows=rs, cols=cs)
            assert np.array_equal(got["status"], want["status"]), (nkeep, daily, got["status"], want["status"])
        ctx.close()
        assert len(np.unique(want["status"])) >= 2                       # (more than one kind of outcome in the window)


def test_krig_tiny_neighbourhoods_in_a_fresh_context(env, orc):
    """Explicit bandwidths of 6 ... 16 stations (below anything the reference's ladder holds; the point entries accept them):
    the pair-distance slab of such a neighbourhood has one block row, the smallest kriging kernel reads three and masks
    only the last itself -- the middle one used to be whatever the allocation held (zeros in a fresh context: 0 x -inf = NaN,
    TWX_CELL_NUMERIC; found by tests/tools/gpu_soak_points.py)."""
    lib, grid = env["lib"], env["grid"]
    cells = np.array([[10, 12], [40, 41], [77, 3], [55, 90]])
    for k in (6, 8, 12, 16, 17, 31):
        ctx = lib.Context()                                              # fresh allocations every time
        ctx.set_stations(lib.TMIN, env["tmin"], with_obs=False)
        pts = _pts(ctx, grid, cells, "tmin")
        mean, var, used, st, _ = ctx.krig_points(lib.TMIN, pts, [1, 4, 8, 12], nnghs=k)
        ctx.close()
        for i in range(4):
            rc, m, v, ku, _ = orc.krig(env["dbn"], env["prm"], orc.make_pt(pts["lon"][i], pts["lat"][i], pts["elev"][i], pts["tdi"][i],
                                                                           pts["lst"][i]), [1, 4, 8, 12][i], k)
            assert st[i] == rc == 0 and used[i] == ku == k
            assert abs(mean[i] - m) < TOL and abs(var[i] - v) < TOL, (k, i, mean[i], m, var[i], v)


def test_api_misuse_is_reported_not_crashed(env):
    lib = env["lib"]
    ctx = lib.Context()
    with pytest.raises(lib.TwxError, match="no station table"):
        ctx.knn(lib.TMIN, [-110.0], [45.0], 10)
    ctx.set_stations(lib.TMIN, env["tmin"], with_obs=False)
    with pytest.raises(lib.TwxError, match="needs observations"):
        ctx.interp_grid(env["grid"], variables=("tmin",), daily=True, rows=slice(0, 2), cols=slice(0, 2))
    with pytest.raises(lib.TwxError, match="bad arguments"):
        ctx.knn(lib.TMIN, [-110.0], [45.0], 400)
    pts = ctx.make_pts(-110.0, 45.0, 1500.0, 30.0, np.zeros(12))
    with pytest.raises(lib.TwxError, match="month"):
        ctx.krig_points(lib.TMIN, pts, 13)
    ctx.close()


def test_krig_every_matrix_size_bucket(env, orc):
    """Explicit bandwidths on both sides of every kernel / template boundary (twx_krig_bucket: one-wave kernels in
    steps of 8 neighbours up to 96 -- bordered form k_ukw<m, 0> for k <= 16 m - 8, border-as-columns form k_ukwz<m>
    above --, two- / four-wave kernels k_uk<7..10> beyond), with full and partial 4-column panels, a pure-nugget
    model and a long-range one."""
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    ks = [7, 8, 9, 31, 32, 33, 39, 40, 41, 47, 48, 49, 55, 56, 57, 63, 64, 65, 71, 72, 73, 79, 80, 81, 87, 88, 89,
          95, 96, 97, 103, 104, 105, 106, 110, 111, 112, 113, 119, 120, 121, 122, 127, 128, 129, 135, 136, 137, 138, 143, 144, 145, 147, 150, 152]
    cells = np.argwhere(np.asarray(grid["mask"]) != 0)[::37][:len(ks)]
    assert len(cells) == len(ks)
    pts = _pts(ctx, grid, cells, "tmin")
    varios = [None, (0.25, 1.4, 60.0), (0.8, 0.0, 0.0), (0.05, 2.0, 900.0)]
    for vi, vario in enumerate(varios):
        mth = 1 + (vi * 5) % 12
        mean, var, used, st, _ = ctx.krig_points(lib.TMIN, pts, mth, nnghs=ks,
                                                 vario=None if vario is None else [vario] * len(ks))
        assert np.all(st == 0), st
        assert np.array_equal(used, ks)
        for i, (r, c) in enumerate(cells):
            pt = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c],
                             grid["lst_night"][:, r, c])
            rc, m, v, u, _ = orc.krig(env["dbn"], env["prm"], pt, mth, nnghs=ks[i], vario=vario)
            assert rc == 0 and u == ks[i]
            assert abs(mean[i] - m) < TOL and abs(var[i] - v) < TOL, (ks[i], vario, mean[i], m, var[i], v)


def test_grid_many_small_batches_equal_one_batch(env, golden_case):
    """Ragged window (37 x 57 cells, a partly masked column range) cut into many device batches (row bands of one
    tile height) gives the same arrays, bit for bit, as one batch: batching, tile candidate lists and the per-cell
    distance cache do not leak across band edges."""
    lib, grid = env["lib"], dict(env["grid"])
    mask = np.array(grid["mask"], copy=True)
    mask[10:14, 20:31] = 0
    grid["mask"] = mask
    rs, cs = slice(3, 40), slice(5, 62)
    outs = []
    for batch in (0, 300):                       # default (one batch) vs bands of a single tile row
        ctx = lib.Context(batch_cells=batch)
        ctx.set_stations(lib.TMIN, golden_case[1])
        ctx.set_stations(lib.TMAX, golden_case[2])
        outs.append(ctx.interp_grid(grid, daily=True, rows=rs, cols=cs))
        ctx.close()
    a, b = outs
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    m = mask[rs, cs] != 0
    assert np.all(a["status"][m] == 0) and np.all(a["status"][~m] == -1)


def test_krig_against_40_digit_arbiter(env, orc):
    """The GPU kriging kernels (fp32 pair distances and exponentials, fp64 bordered Cholesky) against the augmented
    system solved in 40-digit arithmetic (oracle/arbiter.py): one-wave kernel, four-wave kernel and the largest
    bandwidth of the ladder; the residual is the fp32 distance / exponent rounding, far inside 1e-4 degC."""
    from oracle import arbiter
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    c = env["dbn"].cols
    r, q, m = 37, 61, 7
    cell = np.array([[r, q]])
    worst = 0.0
    for k, vario in ((35, (0.3, 0.9, 35.0)), (101, (0.05, 2.0, 900.0)), (147, (0.25, 1.4, 60.0))):
        mean, var, used, st, ngh = ctx.krig_points(lib.TMIN, _pts(ctx, grid, cell, "tmin"), m, nnghs=k, vario=[vario],
                                                   want_idx=True)
        assert st[0] == 0 and used[0] == k
        idx = ngh[0, :k]
        pt = (grid["lon"][q], grid["lat"][r], float(grid["elev"][r, q]), float(grid["lst_night"][m - 1, r, q]))
        am, av = arbiter.uk(c["lon"][idx], c["lat"][idx], c["elev"][idx], c["lst"][m - 1, idx], c["norm"][m - 1, idx], pt, *vario)
        worst = max(worst, abs(mean[0] - am), abs(var[0] - av))
    assert worst < 1e-5, worst


def test_grid_tile_table_equals_point_path(env, golden_case, orc):
    """Grid mode gathers the pair-distance cache per tile from an LDS table of the tile's station pairs (k_tile_dist),
    point mode evaluates it per cell (k_cell_dist): same normals and SE, bit for bit in fp64 terms -- the f4 grid outputs
    are the rounded point outputs -- also where two stations coincide (systems flagged singular by rank, not by pivot)."""
    from topowx_amd import _lib, stationdb as sdb
    ctx, grid = env["ctx"], env["grid"]
    rs, cs = slice(21, 39), slice(58, 77)                         # straddles tile edges
    got = ctx.interp_grid(grid, variables=("tmin",), daily=False, rows=rs, cols=cs)
    cells = np.array([(r, c) for r in range(rs.start, rs.stop) for c in range(cs.start, cs.stop)])
    _, norms, se, st = ctx.interp_points(_lib.TMIN, _pts(ctx, grid, cells, "tmin"), daily=False)
    assert np.all(st == 0) and np.all(got["status"] == 0)
    shp = (rs.stop - rs.start, cs.stop - cs.start)
    assert np.array_equal(got["norm_tmin"], norms.T.reshape(12, *shp).astype(np.float32))
    assert np.array_equal(got["se_tmin"], se.T.reshape(12, *shp).astype(np.float32))
    # a duplicated station location near the window: both paths fail exactly the same cells
    grid0, tmin, _ = golden_case
    dup = sdb.StationDataWrkChk(tmin.stns.copy(), "tmin", tmin.days, tmin.var)
    lo, la = grid["lon"][cs.start + 9], grid["lat"][rs.start + 9]
    j = np.argmin((dup.stns[sdb.LON] - lo) ** 2 + (dup.stns[sdb.LAT] - la) ** 2)
    i = np.argsort((dup.stns[sdb.LON] - lo) ** 2 + (dup.stns[sdb.LAT] - la) ** 2)[40]
    dup.stns[sdb.LON][i], dup.stns[sdb.LAT][i] = dup.stns[sdb.LON][j], dup.stns[sdb.LAT][j]
    c2 = _lib.Context()
    c2.set_stations(_lib.TMIN, dup, with_obs=False)
    g2 = c2.interp_grid(grid, variables=("tmin",), daily=False, rows=rs, cols=cs)
    _, n2, _, s2 = c2.interp_points(_lib.TMIN, _pts(c2, grid, cells, "tmin"), daily=False)
    c2.close()
    assert np.array_equal(g2["status"].ravel(), s2) and (s2 == 4).any()
    # ... and the oracle fails the same cells (coincident stations share the full sill: singular, as gstat reports)
    want = orc.interp_grid(orc.Db(dup), None, env["prm"], grid, daily=False, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(g2["status"], want["status"])
    ok = s2 == 0
    assert np.array_equal(g2["norm_tmin"].reshape(12, -1)[:, ok], n2[ok].T.astype(np.float32))


def test_point_runs_share_candidate_lists(env):
    """Consecutive points at one location with one excluded station share a candidate list (one search of the station
    table per run): the same requests in an order without runs give the same bits, point for point."""
    ctx, lib, grid = env["ctx"], env["lib"], env["grid"]
    rng = np.random.default_rng(3)
    cells = np.column_stack([rng.integers(0, grid["lat"].size, 24), rng.integers(0, grid["lon"].size, 24)])
    base = _pts(ctx, grid, cells, "tmin")
    # station-major: 24 locations x (12 months x 3 bandwidths), excluded station varies inside a location for a third
    pts = np.repeat(base, 36)
    mth = np.tile(np.repeat(np.arange(1, 13), 3), 24)
    nn = np.tile(np.array([35, 60, 0]), 24 * 12)
    excl = np.where(np.arange(pts.size) % 3 == 2, 17, -1).astype(np.int32)
    a = ctx.krig_points(lib.TMIN, pts, mth, nnghs=nn, excl=excl)
    perm = rng.permutation(pts.size)                              # no two neighbours alike (almost surely)
    b = ctx.krig_points(lib.TMIN, pts[perm], mth[perm], nnghs=nn[perm], excl=excl[perm])
    assert np.all(a[3] == 0)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x[perm], y, equal_nan=True)


@pytest.mark.parametrize("which", ["tail31_sparse", "tail40", "norm_years_43"])
def test_fixer_limits_of_the_sparse_path_vs_oracle(env, orc, golden_case, which):
    """``fixer_tail`` and the normals period are public parameters.  A window wider than a wave (tail 40) or a
    normals period longer than the sparse kernel's 40-year table must send every flagged cell through the full
    recompute -- not leave it unfixed: ninvalid == oracle, statuses 0, normals recomputed, daily values within 1 LSB.
    And where the sparse path does run (tail 31: 63 lanes) it gives the full recompute's fixed days bit for bit
    (TWX_FLAG_FIX_FULL forces the latter)."""
    import datetime as dt
    import make_golden
    from topowx_amd import synth
    from topowx_amd.dates import get_days_metadata
    lib, grid = env["lib"], env["grid"]
    if which == "norm_years_43":
        days = get_days_metadata(dt.date(1950, 1, 1), dt.date(1992, 12, 31))
        tmin = synth.make_stations(grid["bbox"], 260, 21, "tmin", days, with_obs=True)
        tmax = make_golden.lowered_tmax(synth.make_stations(grid["bbox"], 260, 21, "tmax", days, with_obs=True))
        tail, years, rs, cs = 15, (1950, 1992), slice(40, 43), slice(50, 54)
    else:
        tmin, tmax = golden_case[1], make_golden.lowered_tmax(golden_case[2])
        tail, years, rs, cs = (31 if which == "tail31_sparse" else 40), (1981, 2010), slice(60, 66), slice(11, 19)
    prm = orc.params(fixer_tail=tail, yr0=years[0], yr1=years[1])
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), prm, grid, daily=True, nthreads=8, rows=rs, cols=cs)
    outs = {}
    for flags in (0, lib.FLAG_FIX_FULL):
        ctx = lib.Context(fixer_tail=tail, norm_years=years, flags=flags)
        ctx.set_stations(lib.TMIN, tmin)
        ctx.set_stations(lib.TMAX, tmax)
        outs[flags] = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
        ctx.close()
    got = outs[0]
    assert want["ninvalid"].max() > 0 and want["ninvalid"].max() <= 256          # the cells the sparse path would take
    assert np.array_equal(got["status"], want["status"]) and np.all(got["status"] == 0)
    assert np.array_equal(got["ninvalid"], want["ninvalid"])
    for k in ("norm_tmin", "norm_tmax"):
        assert np.abs(got[k].astype(np.float64) - want[k]).max() < TOL, k
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(got[k].astype(int) - want[k].astype(int))
        assert dd.max() <= 1 and (dd == 0).mean() > 0.9999, k
        assert (got[k] < 32000).all()                                              # no day left at tmin >= tmax garbage / fill
    assert (got["daily_tmin"].astype(int) < got["daily_tmax"].astype(int)).mean() > 0.999
    full = outs[lib.FLAG_FIX_FULL]
    for k in ("daily_tmin", "daily_tmax", "ninvalid", "status"):
        assert np.array_equal(got[k], full[k]), k                                  # sparse == full, packed values bit for bit
    for k in ("norm_tmin", "norm_tmax"):
        assert np.abs(got[k].astype(np.float64) - full[k]).max() <= 4e-6, k        # f4 normals: last bit of the f8 sum


def test_singular_kriging_before_a_later_selection_failure(env, orc, golden_case):
    """VERDICT r4 weak #1c: a cell whose kriging system is singular in month m (two co-located stations inside the
    neighbourhood) AND whose station selection fails in a later month (database smaller than that month's bandwidth).
    The reference's month loop (interp_tair.py:429-437) meets the singular system first: TWX_CELL_NUMERIC, not the
    selection's TWX_CELL_FEW_STATIONS.  The kriging kernels now run for the months before a selection failure."""
    from topowx_amd import stationdb as sdb
    lib, grid = env["lib"], env["grid"]
    NUMERIC, FEW = 4, 1                                                    # TWX_CELL_NUMERIC, TWX_CELL_FEW_STATIONS (include/twx.h)
    rs, cs = slice(0, 12), slice(0, 12)
    la, lo = grid["lat"][rs].mean(), grid["lon"][cs].mean()
    hit = 0
    for nkeep in (118, 125, 135):
        dbs, plain = [], []
        for src in (golden_case[1], golden_case[2]):
            stns = src.stns[:nkeep].copy()
            var = src.var[:, :nkeep].copy()
            # bandwidths that make the order of the months matter: January..June krige with 35 neighbours (the twins are
            # among them), July..December ask for 147 -- more than the database holds
            for m in range(1, 13):
                stns[sdb.get_optim_varname(m)] = 35.0 if m <= 6 else 147.0
                stns[sdb.get_optim_anom_varname(m)] = 35.0
            plain.append(sdb.StationDataWrkChk(stns.copy(), src.var_name, src.days, var.copy()))
            # a twin of the station nearest to the window's centre: same coordinates, its own id (sorted last) and values
            j = int(np.argmin((stns[sdb.LAT] - la) ** 2 + (stns[sdb.LON] - lo) ** 2))
            twin = stns[j:j + 1].copy()
            twin[sdb.STN_ID] = "S9999999"
            for m in range(1, 13):
                twin[sdb.get_norm_varname(m)] += 0.37
            stns = np.concatenate([stns, twin])
            var = np.concatenate([var, var[:, j:j + 1] + np.float32(0.37)], axis=1)
            dbs.append(sdb.StationDataWrkChk(stns, src.var_name, src.days, var))
        ctx = lib.Context()
        ctx.set_stations(lib.TMIN, dbs[0])
        ctx.set_stations(lib.TMAX, dbs[1])
        for daily in (False, True):
            got = ctx.interp_grid(grid, daily=daily, rows=rs, cols=cs)
            want = orc.interp_grid(orc.Db(dbs[0]), orc.Db(dbs[1]), env["prm"], grid, daily=daily, nthreads=8, rows=rs, cols=cs)
            assert np.array_equal(got["status"], want["status"]), (nkeep, daily, got["status"], want["status"])
            assert np.all(got["norm_tmin"][:, want["status"] != 0] == lib.FILL_F4)
        ctx.close()
        # the scenario itself: cells that fail the SELECTION without the twin (a later month's bandwidth exceeds the database;
        # one more station does not change that) and report the singular system with it
        base = orc.interp_grid(orc.Db(plain[0]), orc.Db(plain[1]), env["prm"], grid, daily=False, nthreads=8, rows=rs, cols=cs)
        hit += int(((base["status"] == FEW) & (want["status"] == NUMERIC)).sum())
    assert hit > 50
