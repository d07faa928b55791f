"""Histogram of the kriging bandwidths k of the headline workload (C2 tile, 10 000 stations, Tmin):
    gpurun -- python tests/tools/gpu_k_hist.py
What decides which matrix-size kernel a system runs in (twx_krig_bucket) and how much padding it carries."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from topowx_amd import _lib, synth

Y = X = 250
grid = synth.make_grid("C2", nrows=Y, ncols=X)
stn = synth.make_stations(grid["bbox"], 10000, synth.CONFIGS["C2"][5], "tmin")
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, stn, with_obs=False)
ctx.interp_grid(grid, variables=("tmin",), daily=False)
ks = ctx.last_bandwidths(_lib.TMIN).ravel()
ks = ks[ks > 0]
h = np.bincount(ks)
out = {int(k): int(c) for k, c in enumerate(h) if c}
print(json.dumps(out))
cum = 0
for k, c in out.items():
    cum += c
    print("%4d %7d  %5.1f %%" % (k, c, 100.0 * cum / ks.size))
