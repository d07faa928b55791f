"""What does deflating the daily outputs ON THE GPU (twx_stream_deflate) buy a streamed run?  8 tiles of 250 x 250 cells x
25 203 days (2 500 stations: quick setup) through driver.interp_tiles_streamed: plain int16 / deflated chunks into a discarding
sink, then into NetCDF-4 tile files (ncio.TileSink: plain files; zlib=True with the host deflating; zlib=True fed by the GPU).
    python3 tests/tools/gpu_deflate_probe.py [tiles] [writer_threads] [quick]   ->  gpurun_out/deflate_probe.json, SUMMARY lines
(quick: only the runs into a discarding sink -- what a profiler run wants)"""
import datetime as dt
import json
import os
import shutil
import sys
import time
import zlib

import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from topowx_amd import _lib, driver, ncio, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402
from topowx_amd.interp import Tiler  # noqa: E402

ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
writers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
quick = len(sys.argv) > 3
days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
grid = synth.make_grid("C2", nrows=250 * max(2, -(-ntiles // 8)), ncols=2000 if ntiles > 8 else 1000)
tmin = synth.make_stations(grid["bbox"], 2500, 1, "tmin", days, with_obs=True)
tmax = synth.make_stations(grid["bbox"], 2500, 1, "tmax", days, with_obs=True)
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin)
ctx.set_stations(_lib.TMAX, tmax)
T = 250
tiles = driver.tile_list(grid["mask"], T, T)[:ntiles]
info = Tiler(grid, T, T, 50, 50, process_tiles=()).build_tile_grid_info()
out = "/dev/shm/twx_deflate_probe"
raw_gb = 2 * days.size * T * T * 2 / 1e9
res = {"tiles": len(tiles), "int16_GB_per_tile": raw_gb, "writer_threads": writers}
seen = {}


def keep_sizes(k, a):
    if "deflated_tmin" in a:
        seen[k] = sum(len(b) for v in ("tmin", "tmax") for b in a["deflated_" + v])


for name, kw in (("int16", {}), ("deflated_on_gpu", {"deflate_chunks": (50, 50)})):
    driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=lambda k, a: None, precision="fast", **kw)     # (sizes the slots)
    for prec in ("fast", "exact"):
        log, ms = {}, []
        _, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=keep_sizes, precision=prec, log=log, tile_ms=ms, **kw)
        res["discard_%s_%s" % (name, prec)] = {"wall_s": secs, "device_ms_mean": log["device_ms_mean"], "copy_ms_mean": log["copy_ms_mean"],
                                               "int16_GBps": raw_gb * len(tiles) / secs}
res["deflate_ms_last_tile"] = ctx.timing()["deflate_ms"]
res["deflated_over_int16"] = sum(seen.values()) / (raw_gb * 1e9 * len(seen))
print("SUMMARY", json.dumps(res), flush=True)
if quick:
    ctx.close()
    sys.exit(0)

# one tile's streams against zlib level 1 / 4 of the same shuffled chunks (what the host would store), and a full check
chk = {}


def check(k, a):
    chk.update({n: ([bytes(b) for b in v] if n.startswith("deflated_") else np.array(v)) for n, v in a.items() if hasattr(v, "shape") or n.startswith("deflated_")})


driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=check, precision="fast", deflate_chunks=(50, 50))
plain = {}
driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=lambda k, a: plain.update({n: np.array(a[n]) for n in ("daily_tmin", "daily_tmax")}),
                             precision="fast")
ok = True
z1 = z4 = gpu = 0
for var in ("tmin", "tmax"):
    back = ncio.TileSink._inflate_tile(chk["deflated_" + var], plain["daily_" + var].shape, 50, 50)
    ok = ok and np.array_equal(back, plain["daily_" + var])
    gpu += sum(len(b) for b in chk["deflated_" + var])
    c = np.ascontiguousarray(plain["daily_" + var][:, :50, :50])
    sh = np.ascontiguousarray(c.reshape(-1).view(np.uint8).reshape(-1, 2).T)
    t0 = time.perf_counter(); z1 += len(zlib.compress(sh, 1)); t1 = time.perf_counter(); z4 += len(zlib.compress(sh, 4)); t2 = time.perf_counter()
    res["host_zlib_MBps_one_core"] = {"level1": c.nbytes / (t1 - t0) / 1e6, "level4": c.nbytes / (t2 - t1) / 1e6}
res["tile_inflates_to_the_int16_values"] = bool(ok)
res["ratio_first_chunk"] = {"gpu_rle": sum(len(chk["deflated_" + v][0]) for v in ("tmin", "tmax")) / (2 * c.nbytes), "zlib1": z1 / (2 * c.nbytes), "zlib4": z4 / (2 * c.nbytes)}
res["ratio_tile_gpu"] = gpu / (raw_gb * 1e9)
print("SUMMARY", json.dumps({k: res[k] for k in ("tile_inflates_to_the_int16_values", "ratio_first_chunk", "ratio_tile_gpu", "host_zlib_MBps_one_core")}), flush=True)

# into NetCDF-4 tile files
for name, skw, dkw, sub in (("netcdf4_plain", dict(zlib=False), {}, tiles), ("netcdf4_deflate_host", dict(zlib=True, complevel=1), {}, tiles[:2]),
                            ("netcdf4_deflate_gpu", dict(zlib=True), {"deflate_chunks": (50, 50)}, tiles)):
    shutil.rmtree(out, ignore_errors=True)
    sink = ncio.TileSink(info, out, days, order=[t[0] for t in sub], ahead=3, prep_threads=4, **skw)
    w = writers if (name != "netcdf4_deflate_host") else 1
    _, secs, _ = driver.interp_tiles_streamed(ctx, grid, sub, T, T, daily=True, sink=sink, precision="fast", writer_threads=w, **dkw)
    sink.close()
    st = dict(sink.stats)
    res[name] = {"tiles": st["tiles"], "wall_s": secs, "int16_GBps": st["int16_bytes"] / secs / 1e9, "on_disk_over_int16": st["disk_bytes"] / st["int16_bytes"],
                 "sink_busy_s": st["total_s"], "writer_threads": w}
    print("SUMMARY", name, json.dumps(res[name]), flush=True)
    if name == "netcdf4_deflate_gpu":                       # read one file back through libhdf5's filter pipeline
        t = ncio.read_tile(sink.writer.fpath(info.get_tile_id(tiles[0][0]), "tmax"), "tmax")
        res["file_read_back_equal"] = bool(np.array_equal(t["daily"], plain["daily_tmax"]))
        print("SUMMARY file_read_back_equal", res["file_read_back_equal"], flush=True)
shutil.rmtree(out, ignore_errors=True)
ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "deflate_probe.json"), "w"), indent=1)
