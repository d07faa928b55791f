"""Every cell of the C2 tile (BASELINE.json configs[1]: 250 x 250 cells, 10 000 stations per variable), normals + SE of
both variables, GPU against the CPU oracle -- the full-size check the suite samples (the oracle needs ~1 minute on the GPU
box's host cores).  With --daily: three years of daily values of both variables with a lowered Tmax (fixer), every cell.
python3 tests/tools/gpu_full_tile_parity.py [--daily] [--f64]  ->  gpurun_out/full_tile_parity[_daily].json"""
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
daily = "--daily" in sys.argv
if daily:
    import datetime as dt
    from topowx_amd import stationdb as sdb
    from topowx_amd.dates import get_days_metadata
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1983, 12, 31))
    grid, tmin, tmax = synth.make_case("C2", with_obs=True, days=days)
    stns = tmax.stns.copy()
    for m in range(1, 13):
        stns[sdb.get_norm_varname(m)] -= 7.5                     # a few per cent of the days with tmin >= tmax
    tmax = sdb.StationDataWrkChk(stns, "tmax", days, tmax.var - np.float32(7.5))
else:
    grid, tmin, tmax = synth.make_case("C2")
f64 = "--f64" in sys.argv                                     # TWX_FLAG_UK_F64_ALL: every kriging system on the fp64 build
ctx = _lib.Context(flags=_lib.FLAG_UK_F64_ALL if f64 else 0)
ctx.set_stations(_lib.TMIN, tmin, with_obs=daily)
ctx.set_stations(_lib.TMAX, tmax, with_obs=daily)
t0 = time.perf_counter()
got = ctx.interp_grid(grid, daily=daily)
t1 = time.perf_counter()
ctx.close()
want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=daily, nthreads=min(256, os.cpu_count() or 8))
t2 = time.perf_counter()
ok = want["status"] == 0
res = {"flags": "TWX_FLAG_UK_F64_ALL" if f64 else "default", "cells": int(grid["mask"].size), "cells_ok": int(ok.sum()), "status_equal": bool(np.array_equal(got["status"], want["status"])),
       "gpu_s_incl_transfers": round(t1 - t0, 3), "oracle_s": round(t2 - t1, 1)}
for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
    d = np.abs(got[k].astype(np.float64) - want[k])[:, ok]
    res[k] = {"max_abs_degC": float(d.max()), "p99.9": float(np.quantile(d, 0.999)), "bit_equal_f4_frac": float((d == 0).mean())}
if daily:
    res["ninvalid_equal"] = bool(np.array_equal(got["ninvalid"], want["ninvalid"]))
    res["cells_with_invalid_days"] = int((want["ninvalid"][ok] > 0).sum())
    res["ninvalid_max"] = int(want["ninvalid"][ok].max())
    for k in ("daily_tmin", "daily_tmax"):
        neq = (got[k] != want[k])[:, ok]
        d = np.abs(got[k].astype(np.int32) - want[k].astype(np.int32))[:, ok]
        res[k] = {"values": int(neq.size), "differ": int(neq.sum()), "flip_rate": float(neq.mean()), "max_diff_LSB": int(d.max())}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", ("full_tile_parity_daily" if daily else "full_tile_parity") + ("_f64.json" if f64 else ".json")), "w"), indent=1)
print(json.dumps(res))
