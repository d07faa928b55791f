"""GPU: the asynchronous grid path -- streamed tiles (twx_stream_*), dense station clusters (candidate lists longer
than the LDS fast path), the step25 chunk loop's resume / per-tile log."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tile_stream_equals_synchronous_entry(golden_case):
    """Six 20 x 25 tiles (one partly masked, one fully masked) through a 2-slot stream: every slot reused three
    times, tiles in flight while earlier ones are read back; results identical to twx_interp_grid, bit for bit."""
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    grid = dict(grid)
    mask = np.array(grid["mask"], copy=True)
    mask[25:33, 30:50] = 0
    mask[40:60, 0:25] = 0
    grid["mask"] = mask
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    tiles = [(slice(r, r + 20), slice(c, c + 25)) for r in (20, 40) for c in (0, 25, 50)]
    want = [ctx.interp_grid(grid, daily=True, rows=rs, cols=cs) for rs, cs in tiles]
    st = ctx.stream(20, 25, daily=True, nslots=2)
    got = [None] * len(tiles)
    for i, (rs, cs) in enumerate(tiles):
        st.submit(i & 1, grid, rs, cs)
        if i >= 1:
            got[i - 1] = {k: (np.array(v) if hasattr(v, "shape") else v) for k, v in st.wait((i - 1) & 1).items()}
    got[-1] = {k: (np.array(v) if hasattr(v, "shape") else v) for k, v in st.wait((len(tiles) - 1) & 1).items()}
    st.close()
    ctx.close()
    for g, w in zip(got, want):
        assert g["device_ms"] > 0
        for k in w:
            assert np.array_equal(g[k], w[k]), k
    assert np.all(got[3]["status"] == -1) and np.all(got[3]["daily_tmin"] == _lib.FILL_I2)      # the fully masked tile


def _cluster_db(base, nclust, var, with_obs, box=(45.55, 45.85, -110.7, -110.4)):
    """``base`` plus ``nclust`` stations inside a 0.3-degree box of the golden grid (ids sort after the base's)."""
    from topowx_amd import stationdb as sdb, synth
    extra = synth.make_stations(box, nclust, 77, var, base.days, with_obs=with_obs, expand_deg=0.0)
    extra.stns[sdb.STN_ID] = ["T%07d" % i for i in range(extra.stns.size)]           # sorts after the 'S...' ids
    stns = np.concatenate([base.stns, extra.stns])
    obs = np.concatenate([base.var, extra.var], axis=1) if with_obs else None
    return sdb.StationDataWrkChk(stns, var, base.days, obs)


@pytest.mark.parametrize("nclust,flags,overflow", [(1500, 0, False), (6000, 0, False), (14000, 0, False), (14000, "nosync", True)])
def test_dense_station_cluster(golden_case, orc, nclust, flags, overflow):
    """A cluster of stations far denser than the tile size: the candidate lists of its tiles outgrow the 512-slot LDS
    path.  Up to 4 096 candidates the full-list kernel ranks them directly; a batch with a longer list is run again with
    lists as long as needed (up to 15 872: the readback that sizes the kriging launches carries the longest list) --
    results = oracle, which searches all stations.  Only where the host may not look (TWX_FLAG_NO_HOST_SYNC) the cells
    of such a tile still fail with TWX_CELL_CAND_OVERFLOW instead of using a truncated list."""
    from topowx_amd import _lib
    grid, tmin, _ = golden_case
    db = _cluster_db(tmin, nclust, "tmin", False)
    ctx = _lib.Context(flags=_lib.FLAG_NO_HOST_SYNC if flags == "nosync" else 0)
    ctx.set_stations(_lib.TMIN, db, with_obs=False)
    rs, cs = slice(30, 46), slice(44, 60)                                             # inside the cluster's box
    got = ctx.interp_grid(grid, variables=("tmin",), rows=rs, cols=cs)
    ctx.close()
    if overflow:
        assert np.all(got["status"] == 7) and np.all(got["norm_tmin"] == _lib.FILL_F4)
        return
    assert np.all(got["status"] == 0)
    want = orc.interp_grid(orc.Db(db), None, orc.params(), grid, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(want["status"], got["status"])
    assert np.abs(got["norm_tmin"].astype(np.float64) - want["norm_tmin"]).max() < 1e-4
    assert np.abs(got["se_tmin"].astype(np.float64) - want["se_tmin"]).max() < 1e-4


def test_cluster_beyond_the_longest_list_still_fails_cleanly(golden_case, orc):
    """17 000 stations inside a 0.1-degree box: the tiles at its centre have more candidates than the re-run can rank
    (15 872).  Their cells fail with TWX_CELL_CAND_OVERFLOW; the other cells of the same batch are computed (== oracle)."""
    from topowx_amd import _lib
    grid, tmin, _ = golden_case
    db = _cluster_db(tmin, 17000, "tmin", False, box=(45.65, 45.75, -110.6, -110.5))
    r = int(np.abs(grid["lat"] - 45.7).argmin()) // 8 * 8
    c = int(np.abs(grid["lon"] + 110.55).argmin()) // 8 * 8
    rs, cs = slice(r - 24, r + 32), slice(c - 24, c + 32)                             # 7 x 7 tiles around the centre
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, db, with_obs=False)
    got = ctx.interp_grid(grid, variables=("tmin",), rows=rs, cols=cs)
    ctx.close()
    st = got["status"]
    assert set(np.unique(st)) == {0, 7} and st[24:32, 24:32].min() == 7               # the central tile, at least
    assert np.all(got["norm_tmin"][:, st == 7] == _lib.FILL_F4)
    want = orc.interp_grid(orc.Db(db), None, orc.params(), grid, nthreads=8, rows=rs, cols=cs)
    assert np.all(want["status"] == 0)                                                 # (the reference has no such limit)
    ok = st == 0
    assert np.abs(got["norm_tmin"].astype(np.float64) - want["norm_tmin"])[:, ok].max() < 1e-4


def test_dense_cluster_daily_through_long_lists(golden_case, orc):
    """Daily output of a batch that was re-run with long candidate lists (no per-tile LDS tables: per-cell pair distances,
    gather sums, fixer lists from the rank-order hat rows): same values as the oracle."""
    import make_golden
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    dbn = _cluster_db(tmin, 14000, "tmin", True)
    dbx = make_golden.lowered_tmax(_cluster_db(tmax, 14000, "tmax", True))          # (a few tmin >= tmax days: the fixer)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, dbn)
    ctx.set_stations(_lib.TMAX, dbx)
    rs, cs = slice(32, 40), slice(44, 60)
    got = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
    ctx.close()
    want = orc.interp_grid(orc.Db(dbn), orc.Db(dbx), orc.params(), grid, daily=True, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(got["status"], want["status"]) and np.all(got["status"] == 0)
    assert np.array_equal(got["ninvalid"], want["ninvalid"]) and want["ninvalid"].max() > 0
    for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
        assert np.abs(got[k].astype(np.float64) - want[k]).max() < 1e-4, k
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(got[k].astype(int) - want[k].astype(int))
        assert dd.max() <= 1 and (dd == 0).mean() > 0.9999


def test_step25_resume_and_tile_log(golden_case, tmp_path):
    """A second run skips the tiles whose output exists (tiling.py:258-275, step25:354-356); an interrupted tile
    (temporary name only) is redone; every tile leaves one JSON record."""
    from topowx_amd import step25
    grid, tmin, tmax = golden_case
    sub = {k: (v[:40, :40] if k in ("mask", "elev", "tdi", "climdiv") else v) for k, v in grid.items()}
    sub["lat"], sub["lon"] = grid["lat"][:40], grid["lon"][:40]
    sub["lst_night"], sub["lst_day"] = grid["lst_night"][:, :40, :40], grid["lst_day"][:, :40, :40]
    lines = []
    out = str(tmp_path)
    stores = step25.proc_work(sub, tmin, tmax, tile_size=20, chunk_size=10, daily=True, out_dir=out, log=lines.append)
    assert stores == {}                                                  # written and dropped, nothing kept in memory
    recs = [json.loads(x) for x in lines]
    assert sorted(r["tile"] for r in recs) == ["h00v00", "h00v01", "h01v00", "h01v01"]
    assert all(r["cells"] == 400 and r["ok"] == 400 and r["failures"] == {} and r["device_ms"] > 0 and r["bytes"] > 0
               for r in recs)
    assert sorted(os.listdir(out)) == ["h00v00.npz", "h00v01.npz", "h01v00.npz", "h01v01.npz"]
    first = {t: dict(np.load(os.path.join(out, t + ".npz"))) for t in ("h00v00", "h01v01")}
    # simulate an interrupted tile and a lost one
    os.replace(os.path.join(out, "h01v01.npz"), os.path.join(out, "h01v01.part.npz"))
    os.remove(os.path.join(out, "h00v00.npz"))
    lines2 = []
    step25.proc_work(sub, tmin, tmax, tile_size=20, chunk_size=10, daily=True, out_dir=out, log=lines2.append)
    log2 = step25.proc_work.last_log
    assert sorted(r["tile"] for r in log2 if r.get("skipped")) == ["h00v01", "h01v00"]
    assert sorted(json.loads(x)["tile"] for x in lines2) == ["h00v00", "h01v01"]
    for t, a in first.items():                                           # redone tiles are identical
        b = np.load(os.path.join(out, t + ".npz"))
        for k in a:
            assert np.array_equal(a[k], b[k]), (t, k)


def test_no_host_sync_mode_equals_default(golden_case):
    """TWX_FLAG_NO_HOST_SYNC (worst-case kriging grids, nothing read back) gives the same bits as the default."""
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    outs = []
    for flags in (0, _lib.FLAG_NO_HOST_SYNC):
        ctx = _lib.Context(flags=flags)
        ctx.set_stations(_lib.TMIN, tmin)
        ctx.set_stations(_lib.TMAX, tmax)
        outs.append(ctx.interp_grid(grid, daily=True, rows=slice(5, 41), cols=slice(50, 97)))
        t = ctx.timing()
        assert t["uk_solves"] == 36 * 47 * 24 and t["uk_launches"] >= 2
        ctx.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_driver_streamed_tiles_equal_synchronous(golden_case, tmp_path):
    """driver.interp_tiles_streamed (three pinned slots, writer thread) == driver.interp_tiles on the same tiles."""
    from topowx_amd import _lib, driver
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    tiles = driver.tile_list(grid["mask"], 25, 25)[:7]
    want = driver.interp_tiles(grid, driver.gpu_compute(ctx, daily=True), tiles, 25, 25)
    seen = []

    def sink(k, arrays):
        np.savez(str(tmp_path / ("t%d.npz" % k)), **{n: v for n, v in arrays.items() if hasattr(v, "shape")})
        seen.append(k)
    log = {}
    res, secs, dev_ms = driver.interp_tiles_streamed(ctx, grid, tiles, 25, 25, daily=True, sink=sink, precision="fast", log=log)
    assert log["precision"] == "fast" and log["tiles_fast"] == 7 and log["tiles_exact"] == 0 and log["copy_ms_mean"] > 0
    # "exact": every tile equals a TWX_FLAG_UK_F64_ALL run; the context is handed back in the fast mode
    exact, _, _ = driver.interp_tiles_streamed(ctx, grid, tiles[:3], 25, 25, daily=True, precision="exact")
    again = driver.interp_tiles(grid, driver.gpu_compute(ctx, daily=True), tiles[:1], 25, 25)
    ctx.close()
    ctx64 = _lib.Context(flags=_lib.FLAG_UK_F64_ALL)
    ctx64.set_stations(_lib.TMIN, tmin)
    ctx64.set_stations(_lib.TMAX, tmax)
    want64 = driver.interp_tiles(grid, driver.gpu_compute(ctx64, daily=True), tiles[:3], 25, 25)
    ctx64.close()
    for k, _, _, _ in tiles[:3]:
        for name in want64[k]:
            assert np.array_equal(exact[k][name], want64[k][name]), (k, name)
    for name in want[tiles[0][0]]:
        assert np.array_equal(again[tiles[0][0]][name], want[tiles[0][0]][name]), name
    assert res is None and seen == [t[0] for t in tiles] and dev_ms > 0 and secs > 0
    for k, _, _, _ in tiles:
        got = np.load(str(tmp_path / ("t%d.npz" % k)))
        for name in want[k]:
            assert np.array_equal(got[name], want[k][name]), (k, name)


def test_a_failing_sink_stops_the_run_and_is_reported(golden_case):
    """The sink's exception reaches the caller, no further tile is interpolated once it is seen, and the context -- handed back
    in the fast mode, its stream closed -- still computes what it computed before."""
    from topowx_amd import _lib, driver
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    tiles = driver.tile_list(grid["mask"], 20, 20)
    want = driver.interp_tiles(grid, driver.gpu_compute(ctx), tiles[:1], 20, 20)
    seen = []

    def sink(k, arrays):
        seen.append(k)
        if len(seen) == 2:
            raise IOError("disk full")
    for writers in (1, 2):
        del seen[:]
        with pytest.raises(IOError, match="disk full"):
            driver.interp_tiles_streamed(ctx, grid, tiles, 20, 20, sink=sink, precision="exact", writer_threads=writers)
        assert 2 <= len(seen) < len(tiles) and sorted(seen) == [t[0] for t in tiles[:len(seen)]]
    again = driver.interp_tiles(grid, driver.gpu_compute(ctx), tiles[:1], 20, 20)
    ctx.close()
    for name in want[tiles[0][0]]:
        assert np.array_equal(again[tiles[0][0]][name], want[tiles[0][0]][name]), name


def test_stream_outlives_context_close(golden_case):
    """twx_destroy closes a context's open streams (they hold device images and pinned blocks of that context); the
    Python TileStream then only forgets its handle -- in either order of close() / garbage collection nothing is freed
    twice or through a dead context."""
    import gc
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    st = ctx.stream(16, 16, daily=False, nslots=2)
    st.submit(0, grid, slice(0, 16), slice(0, 16))
    out = st.wait(0)
    assert np.all(out["status"] == 0)
    ctx.close()                      # destroys the stream too
    st.close()                       # must be a no-op now
    del st
    gc.collect()
    # the other order
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    st = ctx.stream(16, 16, daily=False, nslots=2)
    st.close()
    ctx.close()


def test_non_finite_observations_are_rejected(golden_case):
    """The database is serially complete (station_data.py:547-616); a NaN observation would poison every cell of a tile in
    the table walk of k_daily_tile (0 * NaN), so twx_set_stations refuses the table (include/twx.h)."""
    from topowx_amd import _lib, stationdb as sdb
    _, tmin, _ = golden_case
    obs = np.array(tmin.var, np.float32, copy=True)
    obs[17, 5] = np.nan
    bad = sdb.StationDataWrkChk(tmin.stns.copy(), "tmin", tmin.days, obs)
    ctx = _lib.Context()
    with pytest.raises(_lib.TwxError, match="NaN"):
        ctx.set_stations(_lib.TMIN, bad)
    obs[17, 5] = np.inf
    with pytest.raises(_lib.TwxError, match="NaN"):
        ctx.set_stations(_lib.TMIN, sdb.StationDataWrkChk(tmin.stns.copy(), "tmin", tmin.days, obs))
    ctx.set_stations(_lib.TMIN, tmin)             # the context is still usable
    ctx.close()


def test_streamed_tiles_into_netcdf4_tile_files(golden_case, tmp_path):
    """driver.interp_tiles_streamed with ncio.TileSink as its sink (the reference's workers write every chunk into the tile's
    netCDF, step25:177-185, tiling.py:488-537): the files hold what the synchronous path computes, chunked (ndays, cy, cx) as the
    reference's TileWriter lays them out, in both forms -- chunks copied straight into the file's pages / deflated by the sink."""
    from topowx_amd import _lib, driver, h5nc, ncio
    from topowx_amd.interp import Tiler
    if not h5nc.available():
        pytest.skip("libhdf5 not loadable")
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    tiles = driver.tile_list(grid["mask"], 50, 50)
    want = driver.interp_tiles(grid, driver.gpu_compute(ctx, daily=True), tiles, 50, 50)
    info = Tiler(grid, 50, 50, 10, 10, process_tiles=()).build_tile_grid_info()
    for zl in (False, True):
        out = str(tmp_path / ("tiles%d" % zl))
        sink = ncio.TileSink(info, out, tmin.days, threads=6, zlib=zl, order=[t[0] for t in tiles], verify=(tiles[1][0],))
        # (two sink calls in flight for the uncompressed form: TileSink is thread-safe, driver: writer_threads)
        driver.interp_tiles_streamed(ctx, grid, tiles, 50, 50, daily=True, sink=sink, precision="fast", writer_threads=1 if zl else 2)
        sink.close()
        assert sink.stats["tiles"] == 4 and sink.stats["verified"] == 1
        for k, i, j, _ in tiles:
            for var in ("tmin", "tmax"):
                t = ncio.read_tile(sink.writer.fpath(info.get_tile_id(k), var), var)
                assert np.array_equal(t["daily"], want[k]["daily_" + var]) and np.array_equal(t["norm"], want[k]["norm_" + var])
                assert np.array_equal(t["se"], want[k]["se_" + var]) and np.array_equal(t["ninvalid"], want[k]["ninvalid"])
                np.testing.assert_array_equal(t["lat"], grid["lat"][i:i + 50])
        ds = ncio.open_dataset(sink.writer.fpath(info.get_tile_id(0), "tmax"))
        assert ds.variables["tmax"].chunking() == [tmin.days.size, 10, 10] and ds.variables["tmax"].filters()["zlib"] == zl
        ds.close()
    ctx.close()
