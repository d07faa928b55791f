"""GPU: the daily outputs of a streamed tile as HDF5 shuffle + deflate chunk bytes formed on the device (twx_stream_deflate,
csrc/twx_deflate.h).  The reference reaches that storage form through netCDF4-python's ``zlib=True`` (tiling.py:720,894,913,
1035).  Checks: (1) zlib -- the decoder inside libhdf5 -- inflates every stream to the shuffled chunk of what the synchronous
entry computes; (2) the GPU's bytes equal the CPU restatement's (oracle/deflate_oracle.py), byte for byte; (3) a NetCDF-4 tile
file whose chunks were appended with H5Dwrite_chunk reads back through libhdf5 as the same int16 arrays."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tile_chunks(daily, cy, cx):
    return [np.ascontiguousarray(daily[:, r0:r0 + cy, c0:c0 + cx]) for r0 in range(0, daily.shape[1], cy)
            for c0 in range(0, daily.shape[2], cx)]


@pytest.mark.parametrize("rows,cols,cy,cx", [(slice(20, 60), slice(30, 80), 10, 10),      # 20 chunks of 109 600 values: 7 blocks each
                                             (slice(0, 30), slice(60, 100), 30, 40),      # one chunk: 21 stored blocks, 81 segments (6 of them counted for the code)
                                             (slice(40, 52), slice(8, 22), 4, 7)])        # odd shapes: rows of 7 cells
def test_deflated_chunks_equal_the_restatement_and_inflate_to_the_daily_values(golden_case, rows, cols, cy, cx):
    from oracle import deflate_oracle as dorc
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    grid = dict(grid)
    mask = np.array(grid["mask"], copy=True)
    mask[45:50, 10:40] = 0                                    # fill values inside some chunks
    grid["mask"] = mask
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    want = ctx.interp_grid(grid, daily=True, rows=rows, cols=cols)
    Y, X = want["status"].shape
    st = ctx.stream(Y, X, daily=True, nslots=2, deflate_chunks=(cy, cx))
    st.submit(0, grid, rows, cols)
    st.submit(1, grid, rows, cols)                            # (the same tile in the other device image)
    outs = []
    for slot in (0, 1):
        o = st.wait(slot)
        outs.append({k: ([bytes(b) for b in v] if k.startswith("deflated_") else (np.array(v) if hasattr(v, "shape") else v))
                     for k, v in o.items()})
    dev_ms, copy_ms = st.times(1)
    timing = ctx.timing()
    st.close()
    ctx.close()
    assert dev_ms > 0 and copy_ms > 0 and timing["deflate_ms"] > 0
    nd = tmin.days.size
    for o in outs:
        assert "daily_tmin" not in o and o["deflate_chunks"] == (cy, cx)
        for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax", "ninvalid", "status"):
            assert np.array_equal(o[k], want[k]), k
        for var in ("tmin", "tmax"):
            chunks = _tile_chunks(want["daily_" + var], cy, cx)
            blobs = o["deflated_" + var]
            assert len(blobs) == len(chunks)
            for c, (blob, chunk) in enumerate(zip(blobs, chunks)):
                raw = np.frombuffer(zlib.decompress(blob), np.uint8)          # zlib checks the Adler-32 itself
                lo, hi = dorc.shuffled(chunk)
                assert raw.size == 2 * lo.size and np.array_equal(raw[:lo.size], lo) and np.array_equal(raw[lo.size:], hi), (var, c)
                assert np.array_equal(dorc.inflate_chunk(blob, nd, cy, cx), chunk)
    # byte for byte against the restatement (pure Python: a few chunks)
    o = outs[0]
    for var in ("tmin", "tmax"):
        chunks = _tile_chunks(want["daily_" + var], cy, cx)
        table = dorc.tile_table(want["daily_" + var], cy, cx)         # the variable's Huffman code, from every 16th segment of every chunk
        for c in sorted({0, len(chunks) // 2, len(chunks) - 1}):
            if chunks[c].size > 400000:
                continue
            assert o["deflated_" + var][c] == dorc.deflate_chunk(chunks[c], table), (var, c)
    sizes = [len(b) for b in o["deflated_tmin"]]
    assert max(sizes) <= 2 * nd * cy * cx + 5 * (nd * cy * cx // dorc.SEG + nd * cy * cx // 65535 + 2) + 11       # never longer than stored


def test_streamed_deflated_tiles_into_netcdf4(golden_case, tmp_path):
    """driver.interp_tiles_streamed(deflate_chunks=...) -> ncio.TileSink(zlib=True): the sink appends the GPU's chunk bytes with
    H5Dwrite_chunk; the files read back through libhdf5 (shuffle + deflate filters) as what the synchronous path computes."""
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
    sink = ncio.TileSink(info, str(tmp_path), tmin.days, threads=4, zlib=True, verify=(tiles[2][0],))
    log = {}
    driver.interp_tiles_streamed(ctx, grid, tiles, 50, 50, daily=True, sink=sink, precision="fast", deflate_chunks=(10, 10), log=log)
    sink.close()
    assert sink.stats["tiles"] == 4 and sink.stats["verified"] == 1 and log["tiles_fast"] == 4
    assert sink.stats["int16_bytes"] == 4 * 2 * want[0]["daily_tmin"].nbytes and sink.stats["disk_bytes"] < sink.stats["int16_bytes"]
    for k, i, j, _ in tiles:
        for var in ("tmin", "tmax"):
            t = ncio.read_tile(sink.writer.fpath(info.get_tile_id(k), var), var)
            assert np.array_equal(t["daily"], want[k]["daily_" + var]) and np.array_equal(t["norm"], want[k]["norm_" + var])
            assert np.array_equal(t["ninvalid"], want[k]["ninvalid"])
    ds = ncio.open_dataset(sink.writer.fpath(info.get_tile_id(0), "tmax"))
    f = ds.variables["tmax"].filters()
    assert ds.variables["tmax"].chunking() == [tmin.days.size, 10, 10] and f["zlib"] and f["shuffle"]
    ds.close()
    # the default sink collects the streams; a sink without zlib refuses them
    got, _, _ = driver.interp_tiles_streamed(ctx, grid, tiles[:1], 50, 50, daily=True, precision="fast", deflate_chunks=(25, 50))
    k = tiles[0][0]
    assert len(got[k]["deflated_tmin"]) == 2 and "daily_tmin" not in got[k]
    assert np.array_equal(ncio.TileSink._inflate_tile(got[k]["deflated_tmax"], want[k]["daily_tmax"].shape, 25, 50), want[k]["daily_tmax"])
    plain = ncio.TileSink(info, str(tmp_path / "plain"), tmin.days, threads=2)
    with pytest.raises(IOError, match="zlib=True"):
        driver.interp_tiles_streamed(ctx, grid, tiles[:1], 50, 50, daily=True, sink=plain, precision="fast", deflate_chunks=(10, 10))
    plain.close()
    ctx.close()


def test_deflate_stream_argument_checks(golden_case):
    from topowx_amd import _lib
    grid, tmin, tmax = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    with pytest.raises(_lib.TwxError, match="divide"):
        ctx.stream(20, 20, daily=True, deflate_chunks=(8, 10))
    with pytest.raises(_lib.TwxError, match="daily"):
        ctx.stream(20, 20, daily=False, deflate_chunks=(10, 10))
    st = ctx.stream(20, 20, daily=True, nslots=3, deflate_chunks=(10, 20), variables=("tmax",))
    # three tiles submitted before anything is waited for: the third needs the first one's device image -> the library copies
    # the first one out itself
    for slot, r0 in enumerate((0, 20, 40)):
        st.submit(slot, grid, slice(r0, r0 + 20), slice(10, 30))
    for slot, r0 in enumerate((0, 20, 40)):
        o = st.wait(slot)
        assert "deflated_tmin" not in o and len(o["deflated_tmax"]) == 2
        want = ctx_sync(tmin, tmax, grid, slice(r0, r0 + 20), slice(10, 30))
        got = np.concatenate([np.frombuffer(zlib.decompress(bytes(b)), np.uint8) for b in o["deflated_tmax"]])
        n = tmin.days.size * 10 * 20
        for c in range(2):
            raw = got[c * 2 * n:(c + 1) * 2 * n]
            chunk = np.stack([raw[:n], raw[n:]], axis=1).reshape(-1).view("<i2").reshape(-1, 10, 20)
            assert np.array_equal(chunk, want["daily_tmax"][:, c * 10:(c + 1) * 10, :])
    st.close()
    ctx.close()


def ctx_sync(tmin, tmax, grid, rows, cols):
    from topowx_amd import _lib
    c = _lib.Context()
    c.set_stations(_lib.TMIN, tmin)
    c.set_stations(_lib.TMAX, tmax)
    out = c.interp_grid(grid, daily=True, rows=rows, cols=cols)
    c.close()
    return out
