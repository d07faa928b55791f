#!/usr/bin/env python3
"""One tile of BASELINE.json configs[3]: 250x250 cells, 10 000 stations, 25 203 days (1948-2016),
Tmin + Tmax daily int16 + normals, fixer on.  Spot-checks a 2x2 window against the oracle."""
import datetime as dt
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402


def main():
    days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
    t0 = time.time()
    grid = synth.make_grid("C2")
    tmin = synth.make_stations(grid["bbox"], 10000, 1, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], 10000, 1, "tmax", days, with_obs=True)
    print("synthetic tile + obs (2 x %.2f GB) in %.0f s" % (tmin.var.nbytes / 1e9, time.time() - t0), flush=True)
    ctx = _lib.Context()
    t0 = time.time()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    print("station tables + obs relayout/upload in %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    out = ctx.interp_grid(grid, daily=True)
    t1 = time.time()
    tm = ctx.timing()
    ncd = 250 * 250 * days.size * 2
    print("host call %.1f s (D2H of %.1f GB int16), device %.3f s %s (ms)" % (
        t1 - t0, 2 * out["daily_tmin"].nbytes / 1e9, tm["total_ms"] / 1e3,
        {k: round(v, 1) for k, v in tm.items() if k.endswith("_ms")}))
    print("C4 tile: %.3g cell-days/s on one GPU (device); status %s; cells with fixed days %d, max ninvalid %d" % (
        ncd / (tm["total_ms"] * 1e-3), dict(zip(*np.unique(out["status"], return_counts=True))),
        (out["ninvalid"] > 0).sum(), out["ninvalid"].max()))
    orc.build()
    sl = (slice(131, 133), slice(77, 79))            # straddles the two device batches' row bands? (128-row bands)
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=True, nthreads=8, rows=sl[0], cols=sl[1])
    assert np.array_equal(want["ninvalid"], out["ninvalid"][sl])
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(want[k].astype(int) - out[k][(slice(None),) + sl].astype(int))
        print("   %s vs oracle (2x2 cells x %d days): max LSB diff %d, equal %.6f" % (k, days.size, dd.max(), (dd == 0).mean()))
        assert dd.max() <= 1
    for k in ("norm_tmin", "norm_tmax", "se_tmin"):
        assert np.abs(want[k].astype(np.float64) - out[k][(slice(None),) + sl]).max() < 1e-4
    ctx.close()


if __name__ == "__main__":
    main()
