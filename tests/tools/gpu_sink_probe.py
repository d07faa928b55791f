"""Where does the wall of a streamed run into ncio.TileSink go?  8 tiles of 250 x 250 cells x 25 203 days (2 500 stations: quick
setup) through driver.interp_tiles_streamed into a discarding sink and into TileSink, with the sink's per-tile timeline.
    python3 tests/tools/gpu_sink_probe.py [ahead] [threads] [prep_threads] [unused] [writer_threads]"""
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
populate = (sys.argv[4] != "0") if len(sys.argv) > 4 else False
writers = int(sys.argv[5]) if len(sys.argv) > 5 else 1
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
res = {"ahead": ahead, "threads": threads, "prep_threads": prep_threads, "populate": populate, "writer_threads": writers}
driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=lambda k, a: None, precision="fast")
t0 = time.perf_counter()
_, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=lambda k, a: None, precision="fast")
res["discard"] = {"secs_inside": secs, "wall_outside": time.perf_counter() - t0, "device_ms": dev}
if os.environ.get("TWX_PROBE_READRATE"):
    # how fast can the host READ the pinned slot the outputs arrive in?  32 threads gather (days x 50 x 50) chunks of one tile into
    # WARM anonymous buffers (no page allocation on the destination side), from the pinned views and from an ordinary copy of them
    from concurrent.futures import ThreadPoolExecutor
    bufs = [np.ones((days.size, 50, 50), np.int16) for _ in range(50)]
    rates = {}

    def probe_sink(k, arrays):
        srcs = {"pinned": (arrays["daily_tmin"], arrays["daily_tmax"])}
        srcs["pageable copy"] = tuple(np.array(a) for a in srcs["pinned"])
        jobs = [(v, r0, c0) for v in (0, 1) for r0 in range(0, T, 50) for c0 in range(0, T, 50)]
        with ThreadPoolExecutor(32) as pool:
            for name, pair in srcs.items():
                for rep_ in range(2):
                    t0_ = time.perf_counter()
                    list(pool.map(lambda j: np.copyto(bufs[(j[0] * 25 + j[1] // 50 * 5 + j[2] // 50)], pair[j[0]][:, j[1]:j[1] + 50, j[2]:j[2] + 50]), jobs))
                    rates[name + (" (again)" if rep_ else "")] = round(2 * pair[0].nbytes / (time.perf_counter() - t0_) / 1e9, 2)
    driver.interp_tiles_streamed(ctx, grid, tiles[:1], T, T, daily=True, sink=probe_sink, precision="fast")
    print("SUMMARY gather of one tile into warm anonymous buffers, 32 threads, GB/s:", rates)
for rep in range(2):
    shutil.rmtree(out, ignore_errors=True)
    sink = ncio.TileSink(info, out, days, threads=threads, order=[t[0] for t in tiles], ahead=ahead, prep_threads=prep_threads)
    line = []
    tt0 = time.perf_counter()

    def timed_sink(k, arrays):
        a = time.perf_counter() - tt0
        sink(k, arrays)
        line.append((k, round(a, 3), round(time.perf_counter() - tt0, 3)))
    _, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=timed_sink, precision="fast", writer_threads=writers)
    wall = time.perf_counter() - tt0
    sink.close()
    st = sink.stats
    res["sink_rep%d" % rep] = {"secs_inside": secs, "wall_outside": wall, "GBps": st["int16_bytes"] / secs / 1e9, "sink_calls_enter_exit": str(line),
                               "prepare_wait_s": st["prepare_s"], "copy_s": st["copy_s"], "fallocate_thread_s": st["fallocate_s"]}
shutil.rmtree(out, ignore_errors=True)
ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sink_probe_%d_%d_%d_%d_%d.json" % (ahead, threads, prep_threads, int(populate), writers)), "w"), indent=1)
print("SUMMARY ahead %d threads %d prep_threads %d populate %d writers %d | discard %.2f s" % (ahead, threads, prep_threads, populate, writers, res["discard"]["secs_inside"]))
for k in ("sink_rep0", "sink_rep1"):
    r = res[k]
    print("SUMMARY %s %.2f GB/s  secs %.2f  prep_wait %.2f  copy %.2f  falloc_thread %.2f  %s" % (k, r["GBps"], r["secs_inside"], r["prepare_wait_s"], r["copy_s"], r["fallocate_thread_s"], r["sink_calls_enter_exit"]))
