"""Every cell of the C2 tile (BASELINE.json configs[1]: 250 x 250 cells, 10 000 stations per variable), normals + SE of
both variables, GPU against the CPU oracle -- the full-size check the suite samples (the oracle needs ~1 minute on the GPU
box's host cores).  python3 tests/tools/gpu_full_tile_parity.py  ->  gpurun_out/full_tile_parity.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib, synth  # noqa: E402

orc.build()
grid, tmin, tmax = synth.make_case("C2")
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
t0 = time.perf_counter()
got = ctx.interp_grid(grid)
t1 = time.perf_counter()
ctx.close()
want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, nthreads=min(256, os.cpu_count() or 8))
t2 = time.perf_counter()
ok = want["status"] == 0
res = {"cells": int(grid["mask"].size), "cells_ok": int(ok.sum()), "status_equal": bool(np.array_equal(got["status"], want["status"])),
       "gpu_s_incl_transfers": round(t1 - t0, 3), "oracle_s": round(t2 - t1, 1)}
for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
    d = np.abs(got[k].astype(np.float64) - want[k])[:, ok]
    res[k] = {"max_abs_degC": float(d.max()), "p99.9": float(np.quantile(d, 0.999)), "bit_equal_f4_frac": float((d == 0).mean())}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "full_tile_parity.json"), "w"), indent=1)
print(json.dumps(res))
