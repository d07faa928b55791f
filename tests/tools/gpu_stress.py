#!/usr/bin/env python3
"""Stress shapes: large masked grid (several batches), full 1948-2016 day axis with the fixer."""
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
    orc.build()
    # (1) 1250 x 1500 cells of the C3 grid (blob mask), 12 000 stations per variable, both variables
    t0 = time.time()
    grid = synth.make_grid("C3", nrows=1250, ncols=1500, full_mask=False)
    tmin = synth.make_stations(grid["bbox"], 12000, 2, "tmin")
    tmax = synth.make_stations(grid["bbox"], 12000, 2, "tmax")
    print("synth %.1fs" % (time.time() - t0), flush=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    t0 = time.time()
    got = ctx.interp_grid(grid)
    t1 = time.time()
    tm = ctx.timing()
    nvalid = int((grid["mask"] != 0).sum())
    st = got["status"]
    print("grid 1250x1500: valid %d, host call %.2fs, device %.0f ms (uk %.0f, select %.0f, tile %.0f) -> %.3g cell-months/s"
          % (nvalid, t1 - t0, tm["total_ms"], tm["uk_ms"], tm["select_ms"], tm["tile_cand_ms"],
             nvalid * 24 / (tm["total_ms"] * 1e-3)), flush=True)
    print("   status:", dict(zip(*np.unique(st, return_counts=True))))
    assert np.all((st == -1) == (grid["mask"] == 0)) and np.all(st[grid["mask"] != 0] == 0)
    dbn, dbx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
    worst = 0.0
    rs = np.random.default_rng(5)
    for _ in range(12):
        r, c = int(rs.integers(0, 1248)), int(rs.integers(0, 1498))
        sl = (slice(r, r + 2), slice(c, c + 2))
        want = orc.interp_grid(dbn, dbx, prm, grid, rows=sl[0], cols=sl[1])
        assert np.array_equal(want["status"], got["status"][sl])
        m = want["status"] == 0
        for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
            if m.any():
                worst = max(worst, float(np.abs(want[k][:, m].astype(np.float64) - got[k][(slice(None),) + sl][:, m]).max()))
    print("   12 random 2x2 windows vs oracle: max |d| = %.2e degC" % worst, flush=True)
    assert worst < 1e-4
    ctx.close()

    # (2) full day axis 1948-2016 (25 203 days), 48x48 cells, 1 500 stations, Tmax lowered so the fixer works hard
    days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
    grid = synth.make_grid("C1", nrows=48, ncols=48)
    t0 = time.time()
    tmin = synth.make_stations(grid["bbox"], 1500, 8, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], 1500, 8, "tmax", days, with_obs=True)
    for m in range(1, 13):
        tmax.stns["norm%02d" % m] -= 6.0
    tmax.var -= np.float32(6.0)
    print("synth obs %.1fs  (%d days)" % (time.time() - t0, days.size), flush=True)
    ctx = _lib.Context()
    t0 = time.time()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    print("upload %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    got = ctx.interp_grid(grid, daily=True)
    t1 = time.time()
    tm = ctx.timing()
    ncd = 48 * 48 * days.size * 2
    print("daily 48x48 x %d days x 2: host call %.2fs, device %.0f ms %s -> %.3g cell-days/s"
          % (days.size, t1 - t0, tm["total_ms"], {k: round(v, 1) for k, v in tm.items() if k.endswith("_ms")},
             ncd / (tm["total_ms"] * 1e-3)), flush=True)
    print("   cells with fixed days: %d of %d, ninvalid min/mean/max %d/%.1f/%d" % (
        (got["ninvalid"] > 0).sum(), 48 * 48, got["ninvalid"].min(), got["ninvalid"].mean(), got["ninvalid"].max()))
    sl = (slice(20, 22), slice(30, 32))
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=True, nthreads=8, rows=sl[0], cols=sl[1])
    assert np.array_equal(want["ninvalid"], got["ninvalid"][sl]), (want["ninvalid"], got["ninvalid"][sl])
    for k in ("norm_tmin", "norm_tmax"):
        assert np.abs(want[k].astype(np.float64) - got[k][(slice(None),) + sl]).max() < 1e-4
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(want[k].astype(int) - got[k][(slice(None),) + sl].astype(int))
        print("   %s vs oracle: max LSB diff %d, equal %.6f" % (k, dd.max(), (dd == 0).mean()))
        assert dd.max() <= 1
    ctx.close()


if __name__ == "__main__":
    main()
