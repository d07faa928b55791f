"""Where does the wall of a streamed run into ncio.TileSink go?  8 tiles of 250 x 250 cells x 25 203 days (2 500 stations: quick
setup) through driver.interp_tiles_streamed into a discarding sink and into TileSink, with the sink's per-tile timeline.
    python3 tests/tools/gpu_sink_probe.py [ahead] [threads] [prep_threads]"""
import datetime as dt
import json
import os
import shutil
import sys
import time

import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from topowx_amd import _lib, driver, ncio, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402
from topowx_amd.interp import Tiler  # noqa: E402

ahead = int(sys.argv[1]) if len(sys.argv) > 1 else 2
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
prep_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 2
populate = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
grid = synth.make_grid("C2", nrows=500, ncols=1000)
tmin = synth.make_stations(grid["bbox"], 2500, 1, "tmin", days, with_obs=True)
tmax = synth.make_stations(grid["bbox"], 2500, 1, "tmax", days, with_obs=True)
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin)
ctx.set_stations(_lib.TMAX, tmax)
T = 250
tiles = driver.tile_list(grid["mask"], T, T)
info = Tiler(grid, T, T, 50, 50, process_tiles=()).build_tile_grid_info()
out = "/dev/shm/twx_sink_probe"
res = {"ahead": ahead, "threads": threads, "prep_threads": prep_threads, "populate": populate}
driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=lambda k, a: None, precision="fast")
t0 = time.perf_counter()
_, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=lambda k, a: None, precision="fast")
res["discard"] = {"secs_inside": secs, "wall_outside": time.perf_counter() - t0, "device_ms": dev}
for rep in range(2):
    shutil.rmtree(out, ignore_errors=True)
    sink = ncio.TileSink(info, out, days, threads=threads, order=[t[0] for t in tiles], ahead=ahead, prep_threads=prep_threads, populate=populate)
    line = []
    tt0 = time.perf_counter()

    def timed_sink(k, arrays):
        a = time.perf_counter() - tt0
        sink(k, arrays)
        line.append((k, round(a, 3), round(time.perf_counter() - tt0, 3)))
    _, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=timed_sink, precision="fast")
    wall = time.perf_counter() - tt0
    sink.close()
    st = sink.stats
    res["sink_rep%d" % rep] = {"secs_inside": secs, "wall_outside": wall, "GBps": st["int16_bytes"] / secs / 1e9, "sink_calls_enter_exit": str(line),
                               "prepare_wait_s": st["prepare_s"], "copy_s": st["copy_s"], "fallocate_thread_s": st["fallocate_s"]}
shutil.rmtree(out, ignore_errors=True)
ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sink_probe_%d_%d_%d_%d.json" % (ahead, threads, prep_threads, int(populate))), "w"), indent=1)
print("SUMMARY ahead %d threads %d prep_threads %d populate %d | discard %.2f s" % (ahead, threads, prep_threads, populate, res["discard"]["secs_inside"]))
for k in ("sink_rep0", "sink_rep1"):
    r = res[k]
    print("SUMMARY %s %.2f GB/s  secs %.2f  prep_wait %.2f  copy %.2f  falloc_thread %.2f  %s" % (k, r["GBps"], r["secs_inside"], r["prepare_wait_s"], r["copy_s"], r["fallocate_thread_s"], r["sink_calls_enter_exit"]))
