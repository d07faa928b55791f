"""GPU: BASELINE.json configs at their own SHAPES (round-1 verdict: configs[2] and configs[4] were only exercised by
builder tools), plus the branches no other test reaches.

* configs[2]-shaped: 500 x 1000 cells cut from the CONUS-shaped grid generator (seed-7 blob mask with empty 250^2
  tiles), 12 000 stations per variable, Tmin + Tmax normals, DEFAULT batch size -> several row bands; a window
  equals the full grid bit for bit and 64 sampled cells match the oracle.
* configs[4]-shaped: leave-one-out normals of ALL 10 000 stations (120 000 kriging systems with the station itself
  excluded), sampled stations against the oracle.
* the 64-bit observation addressing of the daily kernel (the path a > 4 GiB matrix takes) against the 32-bit one.
* a singular GWR system (constant TDI neighbourhood) abandons the whole cell, as the reference's worker does.
* create / aggregate / destroy cycles give all device memory back.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4   # degC (north_star)


@pytest.fixture(scope="module")
def c3s():
    from topowx_amd import _lib, synth
    grid = synth.make_grid("C3", nrows=500, ncols=1000, lat_north=41.0, lon_west=-110.0)
    tmin = synth.make_stations(grid["bbox"], 12000, 2, "tmin")
    tmax = synth.make_stations(grid["bbox"], 12000, 2, "tmax")
    ctx = _lib.Context()                       # default batch_cells: 131 072 cells -> 128-row bands
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    full = ctx.interp_grid(grid)
    yield dict(lib=_lib, grid=grid, tmin=tmin, tmax=tmax, ctx=ctx, full=full)
    ctx.close()


def test_c3_shape_masked_multiband_grid(c3s):
    lib, grid, full = c3s["lib"], c3s["grid"], c3s["full"]
    mask = grid["mask"] != 0
    tiles = [int(mask[i:i + 250, j:j + 250].sum()) for i in (0, 250) for j in (0, 250, 500, 750)]
    assert min(tiles) == 0 and 0.4 < mask.mean() < 0.7          # empty tile(s), ~57 % valid
    assert np.all(full["status"][~mask] == -1) and np.all(full["status"][mask] == 0)
    for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
        assert np.all(full[k][:, ~mask] == lib.FILL_F4) and np.isfinite(full[k][:, mask]).all(), k
    assert np.all(full["ninvalid"][mask] == 0) and np.all(full["ninvalid"][~mask] == lib.FILL_I4)


def test_c3_shape_windows_equal_full_grid(c3s):
    """Windows that straddle the 128-row band edges and tile corners of the full run, incl. one inside an empty tile."""
    ctx, grid, full = c3s["ctx"], c3s["grid"], c3s["full"]
    for rs, cs in ((slice(100, 300), slice(200, 520)), (slice(383, 386), slice(0, 1000)), (slice(0, 250), slice(0, 250)),
                   (slice(250, 500), slice(0, 250)), (slice(121, 135), slice(731, 745))):
        win = ctx.interp_grid(grid, rows=rs, cols=cs)
        for k in ("status", "ninvalid", "norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
            assert np.array_equal(win[k], full[k][..., rs, cs]), (k, rs, cs)


def test_c3_shape_sampled_cells_vs_oracle(c3s, orc):
    grid, full = c3s["grid"], c3s["full"]
    dbn, dbx, prm = orc.Db(c3s["tmin"]), orc.Db(c3s["tmax"]), orc.params()
    cells = np.argwhere(grid["mask"] != 0)
    cells = cells[np.random.default_rng(17).choice(len(cells), 64, replace=False)]
    worst = 0.0
    for r, c in cells:
        for v, db, lst in (("tmin", dbn, "lst_night"), ("tmax", dbx, "lst_day")):
            pt = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid[lst][:, r, c])
            rc, _, norms, se = orc.interp(db, prm, pt, daily=False)
            assert rc == 0
            worst = max(worst, np.abs(full["norm_" + v][:, r, c] - norms).max(), np.abs(full["se_" + v][:, r, c] - se).max())
    assert worst < TOL, worst


def test_all_stations_leave_one_out_normals(orc):
    """configs[4] scale: every station of the 10 000-station table is interpolated with itself left out
    (XvalTairOverall.run_interp, optimize.py:579-604; normals part), in one batched call."""
    from topowx_amd import stationdb as sdb, synth
    from topowx_amd.interp import XvalTairOverall
    grid = synth.make_grid("C2")
    stn = synth.make_stations(grid["bbox"], 10000, 1, "tmin")
    xo = XvalTairOverall(stn, "tmin")
    ids = stn.stns[sdb.STN_ID][np.isnan(stn.stns[sdb.BAD])]
    _, norms, se, st = xo.run_interp_many(ids, daily=False, raise_on_error=False)
    xo.close()
    assert norms.shape == (ids.size, 12)
    # stations at the rim of the station cloud may lack 148 neighbours; everything else must be solved
    assert (st == 0).mean() > 0.99 and set(np.unique(st)) <= {0, 1}
    db, prm = orc.Db(stn), orc.params()
    c = db.cols
    worst = 0.0
    for j in np.random.default_rng(5).choice(ids.size, 48, replace=False):
        pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
        rc, _, wn, ws = orc.interp(db, prm, pt, excl=int(j), rm_zero_dist=True, daily=False)
        assert rc == st[j]
        if rc == 0:
            worst = max(worst, np.abs(norms[j] - wn).max(), np.abs(se[j] - ws).max())
    assert worst < TOL, worst
    # leave-one-out error is a real prediction error, not an exact-interpolation artefact
    err = norms[st == 0] - np.column_stack([stn.stns[sdb.get_norm_varname(m)] for m in range(1, 13)])[st == 0]
    assert 0.05 < np.abs(err).mean() < 3.0


def test_daily_64bit_obs_addressing_equals_32bit(golden_case, orc):
    """The three ways a daily value is formed give the SAME bits: every daily sum runs in ascending station-index order
    (GwrWs.perm; a table walk adds the unused rows with weight 0, i.e. nothing) --
    rows of the tile-month staged in LDS (default: hat rows delivered in table order by k_gwr_z_cell),
    gathered from global memory with 32-bit offsets (a tile-month with too many distinct rows; TWX_FLAG_DAILY_GATHER)
    and with 64-bit offsets (stations x days >= 2^30; TWX_FLAG_OBS_ADDR64); the fixer recomputes with the same lists.
    Which path a tile-month takes (<= 208 union rows, i.e. tiling and station density) does not show in any output."""
    from topowx_amd import _lib
    import make_golden
    grid, tmin, tmax = golden_case
    tmax = make_golden.lowered_tmax(tmax)                       # so that the fixer runs too
    rs, cs = slice(40, 75), slice(3, 70)
    outs = []
    for flags in (0, _lib.FLAG_DAILY_GATHER, _lib.FLAG_OBS_ADDR64):    # LDS table / 32-bit gather / 64-bit gather
        ctx = _lib.Context(flags=flags)
        ctx.set_stations(_lib.TMIN, tmin)
        ctx.set_stations(_lib.TMAX, tmax)
        outs.append(ctx.interp_grid(grid, daily=True, rows=rs, cols=cs))
        ctx.close()
    a, g32, g64 = outs
    assert np.all(a["status"] == 0) and a["ninvalid"].max() > 0
    for other in (g32, g64):
        for k in a:                                              # every output on ALL cells: packed days, ninvalid, normals, SE, status
            assert np.array_equal(a[k], other[k]), k
    # ... and both are within 1 LSB of the oracle on the same window (the oracle subtracts the normals term by term)
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=True, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(a["status"], want["status"])
    dn = np.abs(a["ninvalid"].astype(np.int64) - want["ninvalid"])
    assert (dn != 0).sum() <= 3, int((dn != 0).sum())            # a day within an ulp of tmin == tmax may be flagged by one side only
    ok = dn == 0
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(a[k].astype(np.int32) - want[k].astype(np.int32))[:, ok]
        assert dd.max() <= 1 and (dd != 0).mean() < 1e-3, (k, int(dd.max()), float((dd != 0).mean()))


def test_singular_gwr_abandons_the_cell(golden_case, orc):
    """A neighbourhood whose TDI column is constant makes X'WX singular: np.linalg.inv raises in _gwr_series
    (interp_tair.py:1139), interp_pt fails and the worker leaves EVERY output of the cell at fill (step25:154-160)
    -- normals and SE included, although kriging itself (no TDI term) succeeds."""
    from topowx_amd import _lib, stationdb as sdb
    grid, tmin, tmax = golden_case
    flat = sdb.StationDataWrkChk(tmin.stns.copy(), "tmin", tmin.days, tmin.var)
    flat.stns[sdb.TDI] = 0.0
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, flat)
    ctx.set_stations(_lib.TMAX, tmax)
    rs, cs = slice(10, 14), slice(20, 25)
    got = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
    norms_only = ctx.interp_grid(grid, daily=False, rows=rs, cols=cs)
    ctx.close()
    want = orc.interp_grid(orc.Db(flat), orc.Db(tmax), orc.params(), grid, daily=True, nthreads=4, rows=rs, cols=cs)
    assert np.all(want["status"] == 4) and np.array_equal(got["status"], want["status"])
    for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
        assert np.all(got[k] == _lib.FILL_F4), k
    assert np.all(got["daily_tmin"] == _lib.FILL_I2) and np.all(got["daily_tmax"] == _lib.FILL_I2)
    assert np.all(got["ninvalid"] == _lib.FILL_I4)
    # without daily output no GWR is formed: the kriged normals are fine
    assert np.all(norms_only["status"] == 0) and np.isfinite(norms_only["norm_tmin"]).all()


def test_destroy_returns_all_device_memory():
    """twx_destroy releases every buffer, incl. the aggregation / sampling scratch (a per-year step27 loop creates
    and closes one context per call)."""
    import datetime as dt
    import torch
    from topowx_amd import _lib
    from topowx_amd.dates import get_days_metadata
    days = get_days_metadata(dt.date(2001, 1, 1), dt.date(2001, 12, 31))
    raw = np.random.default_rng(0).integers(-3000, 3000, (days.size, 200, 300)).astype(np.int16)
    torch.cuda.init()
    free = []
    for it in range(4):
        ctx = _lib.Context()
        ctx.set_days(days)
        ctx.aggregate(raw, mthly=True, mthly_i16=True, ann=True)
        ctx.sample_points(np.linspace(-110, -109, 300), np.linspace(45, 44, 200), raw[0].astype(np.float32),
                          [-109.5], [44.5])
        ctx.close()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert free[-1] >= free[0] - (8 << 20), free                # no growth from cycle to cycle


def test_single_variable_daily_equals_two_variable_run(golden_case):
    """A one-variable daily request runs the strip kernel (k_daily_grid: gathers through perm), the two-variable request
    the tile kernel (k_daily_tile: table walk): same normals AND same packed days, bit for bit (one summation order:
    ascending station index), where the fixer has nothing to do."""
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    rs, cs = slice(13, 40), slice(61, 99)
    both = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
    one_n = ctx.interp_grid(grid, variables=("tmin",), daily=True, rows=rs, cols=cs)
    one_x = ctx.interp_grid(grid, variables=("tmax",), daily=True, rows=rs, cols=cs)
    ctx.close()
    assert np.all(both["status"] == 0) and both["ninvalid"].max() == 0
    for one, k in ((one_n, "daily_tmin"), (one_x, "daily_tmax")):
        assert np.array_equal(one[k], both[k]), k
    assert np.array_equal(one_n["norm_tmin"], both["norm_tmin"]) and np.array_equal(one_x["se_tmax"], both["se_tmax"])


def test_packed_cluster_union_of_thousands(golden_case):
    """4 000 stations packed into the footprint of HALF a tile (4 x 8 cells, ~30 km2) with 148 neighbours everywhere:
    neighbouring cells then share few neighbours, and the stations the 32 cells of the half-tile krige with number
    thousands -- more than any shared LDS staging could hold (rounds 3-4 staged the union's trigonometry in the pair
    table's space: room for 3 289).  k_tile_dist's per-element path stages per wave, by rank, whatever the union.
    Grid path == point path (k_cell_dist) bit for bit."""
    from topowx_amd import _lib, stationdb as sdb, synth
    grid, _, _ = golden_case
    stn = synth.make_stations(grid["bbox"], 4050, 5, "tmin", expand_deg=0.3)
    rng = np.random.default_rng(9)
    lat, lon = np.asarray(grid["lat"]), np.asarray(grid["lon"])
    dlat, dlon = abs(lat[1] - lat[0]), abs(lon[1] - lon[0])
    n = 4000
    stn.stns[sdb.LAT][:n] = rng.uniform(min(lat[32], lat[35]) - dlat / 2, max(lat[32], lat[35]) + dlat / 2, n)
    stn.stns[sdb.LON][:n] = rng.uniform(min(lon[40], lon[47]) - dlon / 2, max(lon[40], lon[47]) + dlon / 2, n)
    for m in range(1, 13):
        stn.stns[sdb.get_optim_varname(m)] = 148.0
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)
    rs, cs = slice(32, 40), slice(40, 48)
    got = ctx.interp_grid(grid, variables=("tmin",), daily=False, rows=rs, cols=cs)
    kk = ctx.last_bandwidths(_lib.TMIN)
    cells = np.array([(r, c) for r in range(rs.start, rs.stop) for c in range(cs.start, cs.stop)])
    pts = ctx.make_pts(lon[cells[:, 1]], lat[cells[:, 0]], grid["elev"][cells[:, 0], cells[:, 1]],
                       grid["tdi"][cells[:, 0], cells[:, 1]], grid["lst_night"][:, cells[:, 0], cells[:, 1]].T)
    idx, _, _, st_k = ctx.knn(_lib.TMIN, lon[cells[:32, 1]], lat[cells[:32, 0]], 148)
    _, norms, se, st = ctx.interp_points(_lib.TMIN, pts, daily=False)
    ctx.close()
    assert np.all(st_k == 0) and np.unique(idx).size > 3289, np.unique(idx).size   # the half-tile's union
    assert np.array_equal(got["status"].ravel(), st)
    ok = st == 0
    assert ok.sum() >= 32 and np.all(kk[kk > 0] == 148)
    assert np.array_equal(got["norm_tmin"].reshape(12, -1)[:, ok], norms[ok].T.astype(np.float32))
    assert np.array_equal(got["se_tmin"].reshape(12, -1)[:, ok], se[ok].T.astype(np.float32))


def test_dense_stations_large_tile_unions(golden_case, orc):
    """k_tile_dist's second path: a tile whose cells krige with more than 256 distinct stations does not fit the LDS pair
    table and evaluates the distance formula per element instead.  30 000 stations around the 100 x 100 grid (1.4 per
    cell; more than k_tile_cand keeps in LDS, too) with 148 neighbours everywhere give unions of ~400 per 8x8 tile.
    Grid path == point path (per-cell k_cell_dist) bit for bit, sampled cells vs the oracle."""
    from topowx_amd import _lib, stationdb as sdb, synth
    grid, _, _ = golden_case
    stn = synth.make_stations(grid["bbox"], 30000, 11, "tmin", expand_deg=0.3)
    for m in range(1, 13):
        stn.stns[sdb.get_optim_varname(m)] = 148.0
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)
    rs, cs = slice(30, 50), slice(40, 64)
    got = ctx.interp_grid(grid, variables=("tmin",), daily=False, rows=rs, cols=cs)
    kk = ctx.last_bandwidths(_lib.TMIN)
    assert np.all(got["status"] == 0) and np.all(kk[kk > 0] == 148)
    cells = np.array([(r, c) for r in range(rs.start, rs.stop) for c in range(cs.start, cs.stop)])
    pts = ctx.make_pts(grid["lon"][cells[:, 1]], grid["lat"][cells[:, 0]], grid["elev"][cells[:, 0], cells[:, 1]],
                       grid["tdi"][cells[:, 0], cells[:, 1]], grid["lst_night"][:, cells[:, 0], cells[:, 1]].T)
    _, norms, se, st = ctx.interp_points(_lib.TMIN, pts, daily=False)
    ctx.close()
    shp = (rs.stop - rs.start, cs.stop - cs.start)
    assert np.all(st == 0)
    assert np.array_equal(got["norm_tmin"], norms.T.reshape(12, *shp).astype(np.float32))
    assert np.array_equal(got["se_tmin"], se.T.reshape(12, *shp).astype(np.float32))
    db, prm = orc.Db(stn), orc.params()
    worst = 0.0
    for r, c in cells[np.random.default_rng(2).choice(len(cells), 12, replace=False)]:
        pt = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c])
        rc, _, wn, ws = orc.interp(db, prm, pt, daily=False)
        assert rc == 0
        worst = max(worst, np.abs(got["norm_tmin"][:, r - rs.start, c - cs.start] - wn).max(),
                    np.abs(got["se_tmin"][:, r - rs.start, c - cs.start] - ws).max())
    assert worst < TOL, worst
