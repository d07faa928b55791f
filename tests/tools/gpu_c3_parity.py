"""BASELINE.json configs[2] against the CPU oracle on whole tiles of the FULL grid: the 3250 x 7000 CONUS-shaped 30-arcsec
grid (12 963 935 valid cells), 12 000 stations per variable, 12 monthly Tmin + Tmax normals + SE through one
twx_interp_grid call; then N of its 250 x 250 tiles -- the emptiest, the quartiles and the fullest by valid cells, plus
random ones -- through the oracle (every cell of those tiles, 24 normals + 24 SE each).  The suite checks a 500 x 1000 cut.
    python3 tests/tools/gpu_c3_parity.py [ntiles = 8]  ->  gpurun_out/c3_parity.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib, synth  # noqa: E402

ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orc.build()
grid = synth.make_grid("C3")
tmin = synth.make_stations(grid["bbox"], 12000, 2, "tmin")
tmax = synth.make_stations(grid["bbox"], 12000, 2, "tmax")
mask = np.asarray(grid["mask"]) != 0
Y, X = mask.shape
T = 250
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
t0 = time.perf_counter()
got = ctx.interp_grid(grid, daily=False)
gpu_s = time.perf_counter() - t0
ctx.close()
tiles = [(r, c, int(mask[r:r + T, c:c + T].sum())) for r in range(0, Y, T) for c in range(0, X, T)]
tiles = sorted([t for t in tiles if t[2] > 0], key=lambda t: t[2])
pick = [tiles[0], tiles[len(tiles) // 4], tiles[len(tiles) // 2], tiles[3 * len(tiles) // 4], tiles[-1]]
rng = np.random.default_rng(3)
while len(pick) < ntiles:
    t = tiles[int(rng.integers(0, len(tiles)))]
    if t not in pick:
        pick.append(t)
pick = pick[:ntiles]
odn, odx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
res = {"grid": [int(Y), int(X)], "cells_valid": int(mask.sum()), "cells_ok": int((got["status"] == 0).sum()),
       "gpu_s_incl_transfers": round(gpu_s, 2), "tiles_with_valid_cells": len(tiles), "tiles_checked": []}
worst = {k: 0.0 for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax")}
cells = 0
t0 = time.perf_counter()
for r, c, nv in pick:
    rs, cs = slice(r, min(r + T, Y)), slice(c, min(c + T, X))
    want = orc.interp_grid(odn, odx, prm, grid, daily=False, nthreads=min(256, os.cpu_count() or 8), rows=rs, cols=cs)
    ok = want["status"] == 0
    rec = {"row0": r, "col0": c, "valid_cells": nv, "status_equal": bool(np.array_equal(got["status"][rs, cs], want["status"]))}
    for k in worst:
        d = float(np.abs(got[k][:, rs, cs].astype(np.float64) - want[k])[:, ok].max()) if ok.any() else 0.0
        rec[k] = d
        worst[k] = max(worst[k], d)
    cells += int(ok.sum())
    res["tiles_checked"].append(rec)
res["oracle_s"] = round(time.perf_counter() - t0, 1)
res["cells_checked"] = cells
res["values_checked"] = cells * 48
res["status_equal"] = all(t["status_equal"] for t in res["tiles_checked"])
res["max_abs_degC"] = worst
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "c3_parity.json"), "w"), indent=1)
print(json.dumps(res))
