"""Every cell of the C2 tile (BASELINE.json configs[1]: 250 x 250 cells, 10 000 stations per variable), normals + SE of
both variables, GPU against the CPU oracle -- the full-size check the suite samples (the oracle needs ~1 minute on the GPU
box's host cores).  With --daily: three years of daily values of both variables with a lowered Tmax (fixer), every cell.
--years N --year0 Y: the day axis of the daily run (default 3 years from 1981; 69 from 1948 = configs[3]'s 25 203 days:
3.15e9 packed values per tile, ~15 GB of host memory, ~5 minutes of oracle time on 256 threads).
python3 tests/tools/gpu_full_tile_parity.py [--daily [--years N --year0 Y]] [--f64] [--no-guard]  ->  gpurun_out/full_tile_parity[_daily][_Ny].json"""
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
WIN = None                                                    # (rows, cols) window of the grid, or the whole grid
daily = "--daily" in sys.argv
if daily:
    import datetime as dt
    from topowx_amd import stationdb as sdb
    from topowx_amd.dates import get_days_metadata
    nyears = int(sys.argv[sys.argv.index("--years") + 1]) if "--years" in sys.argv else 3
    year0 = int(sys.argv[sys.argv.index("--year0") + 1]) if "--year0" in sys.argv else 1981
    days = get_days_metadata(dt.date(year0, 1, 1), dt.date(year0 + nyears - 1, 12, 31))
    if "--c3" in sys.argv:                                      # a 250 x 250 tile of the FULL configs[2..3] grid, 12 000-station seed-2 tables
        r0, c0 = int(sys.argv[sys.argv.index("--c3") + 1]), int(sys.argv[sys.argv.index("--c3") + 2])
        grid = synth.make_grid("C3")
        tmin = synth.make_stations(grid["bbox"], 12000, 2, "tmin", days, with_obs=True)
        tmax = synth.make_stations(grid["bbox"], 12000, 2, "tmax", days, with_obs=True)
        WIN = (slice(r0, r0 + 250), slice(c0, c0 + 250))
    else:
        grid, tmin, tmax = synth.make_case("C2", with_obs=True, days=days)
    stns = tmax.stns.copy()
    for m in range(1, 13):
        stns[sdb.get_norm_varname(m)] -= 7.5                     # a few per cent of the days with tmin >= tmax
    tmax = sdb.StationDataWrkChk(stns, "tmax", days, tmax.var - np.float32(7.5))
else:
    grid, tmin, tmax = synth.make_case("C2")
f64 = "--f64" in sys.argv                                     # TWX_FLAG_UK_F64_ALL: every kriging system on the fp64 build
noguard = "--no-guard" in sys.argv
ctx = _lib.Context(flags=(_lib.FLAG_UK_F64_ALL if f64 else 0) | (_lib.FLAG_NO_TIE_GUARD if noguard else 0))
ctx.set_stations(_lib.TMIN, tmin, with_obs=daily)
ctx.set_stations(_lib.TMAX, tmax, with_obs=daily)
t0 = time.perf_counter()
kw = {} if WIN is None else {"rows": WIN[0], "cols": WIN[1]}
got = ctx.interp_grid(grid, daily=daily, **kw)
t1 = time.perf_counter()
timing = ctx.timing()
ctx.close()
want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=daily, nthreads=min(256, os.cpu_count() or 8), **kw)
t2 = time.perf_counter()
ok = want["status"] == 0
# No cell is set aside: since round 6 the library itself guards the one discontinuity of the path -- tmin_tmax_fixer's test
# tmin >= tmax (interp_tair.py:170) -- by kriging every cell that has a day with |Tmax - Tmin| < 2e-5 degC a second time on the
# fp64 covariance build (include/twx.h: TWX_FLAG_NO_TIE_GUARD; --no-guard here shows what it catches).
res = {"flags": ("TWX_FLAG_UK_F64_ALL" if f64 else "default") + (" | TWX_FLAG_NO_TIE_GUARD" if noguard else ""), "grid": "C2 tile" if WIN is None else "C3 grid, tile at row %d col %d, 12 000 stations" % (WIN[0].start, WIN[1].start),
       "cells": int(want["status"].size), "cells_ok": int(ok.sum()), "status_equal": bool(np.array_equal(got["status"], want["status"])),
       "gpu_s_incl_transfers": round(t1 - t0, 3), "oracle_s": round(t2 - t1, 1)}
for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
    d = np.abs(got[k].astype(np.float64) - want[k])[:, ok]
    res[k] = {"max_abs_degC": float(d.max()), "p99.9": float(np.quantile(d, 0.999)), "bit_equal_f4_frac": float((d == 0).mean())}
if daily:
    res["ninvalid_equal"] = bool(np.array_equal(got["ninvalid"][ok], want["ninvalid"][ok]))
    res["cells_with_other_ninvalid"] = int((got["ninvalid"][ok] != want["ninvalid"][ok]).sum())
    res["tie_guard"] = {k: timing[k] for k in ("tie_cells", "tie_solves", "tie_ms")}
    res["cells_with_invalid_days"] = int((want["ninvalid"][ok] > 0).sum())
    res["ninvalid_max"] = int(want["ninvalid"][ok].max())
    res["days"] = int(days.size)
    res["day_axis"] = "%d-01-01 .. %d-12-31" % (year0, year0 + nyears - 1)
    for k in ("daily_tmin", "daily_tmax"):
        nval = ndiff = mx = 0
        for d0 in range(0, got[k].shape[0], 512):             # in blocks of days: the full axis is 3 GB per array
            a, b = got[k][d0:d0 + 512][:, ok], want[k][d0:d0 + 512][:, ok]
            neq = a != b
            nval += int(neq.size); ndiff += int(neq.sum())
            if neq.any():
                mx = max(mx, int(np.abs(a[neq].astype(np.int32) - b[neq].astype(np.int32)).max()))
        res[k] = {"values": nval, "differ": ndiff, "flip_rate": ndiff / max(nval, 1), "max_diff_LSB": mx}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = ("full_tile_parity_daily" + ("_%dy" % nyears if nyears != 3 else "")) if daily else "full_tile_parity"
if WIN is not None:
    tag += "_c3"
if noguard:
    tag += "_noguard"
json.dump(res, open(os.path.join(ROOT, "gpurun_out", tag + ("_f64.json" if f64 else ".json")), "w"), indent=1)
print(json.dumps(res))
