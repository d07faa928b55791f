"""Where does the host spend a tile's period in a streamed run?  N full tiles of configs[3]'s shape (250 x 250 cells x 25 203 days,
2 500 stations) through driver.interp_tiles_streamed, plain and with deflate_chunks, with the submit / wait intervals of the
last tiles printed.   python3 tests/tools/gpu_stream_trace.py [tiles]"""
import datetime as dt
import json
import os
import sys

import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from topowx_amd import _lib, driver, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
grid = synth.make_grid("C2", nrows=500, ncols=1000)
tmin = synth.make_stations(grid["bbox"], 2500, 1, "tmin", days, with_obs=True)
tmax = synth.make_stations(grid["bbox"], 2500, 1, "tmax", days, with_obs=True)
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin)
ctx.set_stations(_lib.TMAX, tmax)
T = 250
tiles = driver.tile_list(grid["mask"], T, T)
tiles = [(q,) + tiles[q % len(tiles)][1:] for q in range(n)]          # the same 8 tiles over and over, numbered 0 .. n - 1
for name, kw in (("int16", {}), ("deflated", {"deflate_chunks": (50, 50)})):
    for prec in ("exact",):
        tr, log = [], {}
        _, secs, dev = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=lambda k, a: None, precision=prec, log=log, trace=tr, **kw)
        print("SUMMARY", name, prec, json.dumps({"wall_s": round(secs, 3), "ms_per_tile": round(secs / n * 1e3, 1), "device_ms_mean": round(log["device_ms_mean"], 1),
                                                 "copy_ms_mean": round(log["copy_ms_mean"], 1)}), flush=True)
        for k, what, a, b in tr[:14] + tr[-4:]:
            print("SUMMARY   tile %3d %-6s %8.1f .. %8.1f ms  (%5.1f)" % (k, what, a * 1e3, b * 1e3, (b - a) * 1e3), flush=True)
ctx.close()
