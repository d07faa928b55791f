#!/usr/bin/env python3
"""Larger-scale GPU checks: masked multi-band grid vs oracle samples, daily throughput."""
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
    # ---- (1) C3-shaped sub-extent, blob mask, both variables, several row bands --------------
    grid = synth.make_grid("C3", nrows=420, ncols=900, full_mask=False)
    tmin = synth.make_stations(grid["bbox"], 3000, 2, "tmin")
    tmax = synth.make_stations(grid["bbox"], 3000, 2, "tmax")
    ctx = _lib.Context(batch_cells=100000)          # forces 4 row bands
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    t0 = time.time()
    got = ctx.interp_grid(grid)
    t1 = time.time()
    tm = ctx.timing()
    nvalid = int((grid["mask"] != 0).sum())
    print("masked grid 420x900: valid %d (%.0f%%), host call %.2fs, device %.1f ms, uk %.1f ms -> %.3g cell-months/s (device)"
          % (nvalid, 100.0 * nvalid / grid["mask"].size, t1 - t0, tm["total_ms"], tm["uk_ms"],
             nvalid * 24 / (tm["total_ms"] * 1e-3)))
    st = got["status"]
    print("   status counts:", dict(zip(*np.unique(st, return_counts=True))))
    assert np.all((st == -1) == (grid["mask"] == 0))
    dbn, dbx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
    worst = 0.0
    for (r, c) in [(0, 0), (104, 37), (105, 38), (209, 450), (210, 451), (300, 899), (419, 899), (211, 0), (333, 777)]:
        rs, cs = slice(r, min(420, r + 3)), slice(c, min(900, c + 3))
        want = orc.interp_grid(dbn, dbx, prm, grid, rows=rs, cols=cs)
        assert np.array_equal(want["status"], got["status"][rs, cs]), (r, c)
        m = want["status"] == 0
        for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
            if m.any():
                worst = max(worst, float(np.abs(want[k][:, m].astype(np.float64) - got[k][:, rs, cs][:, m]).max()))
    print("   sampled windows across band / tile edges: max |d| = %.2e degC" % worst)
    assert worst < 1e-4
    ctx.close()

    # ---- (2) daily: 96x96 cells, 600 stations, 1980-1990 (4018 days), both variables ---------------
    days = get_days_metadata(dt.date(1980, 1, 1), dt.date(1990, 12, 31))
    grid = synth.make_grid("C1", nrows=96, ncols=96)
    tmin = synth.make_stations(grid["bbox"], 600, 4, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], 600, 4, "tmax", days, with_obs=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    t0 = time.time()
    got = ctx.interp_grid(grid, daily=True)
    t1 = time.time()
    tm = ctx.timing()
    ncd = 96 * 96 * days.size * 2
    print("daily 96x96 x %d days x 2 vars: host call %.2fs, device %.1f ms (%s) -> %.3g cell-days/s (device)"
          % (days.size, t1 - t0, tm["total_ms"], {k: round(v, 1) for k, v in tm.items() if k.endswith("_ms")},
             ncd / (tm["total_ms"] * 1e-3)))
    print("   cells with fixed days: %d of %d, max ninvalid %d" % ((got["ninvalid"] > 0).sum(), 96 * 96, got["ninvalid"].max()))
    rs, cs = slice(40, 44), slice(50, 54)
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, daily=True, nthreads=8, rows=rs, cols=cs)
    assert np.array_equal(want["ninvalid"], got["ninvalid"][rs, cs])
    for k in ("daily_tmin", "daily_tmax"):
        dd = np.abs(want[k].astype(int) - got[k][:, rs, cs].astype(int))
        print("   %s vs oracle: max LSB diff %d, equal %.6f" % (k, dd.max(), (dd == 0).mean()))
        assert dd.max() <= 1
    ctx.close()

    # ---- (3) config 5 shape: leave-one-out xval of the normals over all stations (step24 path) -----
    grid = synth.make_grid("C2")
    stn = synth.make_stations(grid["bbox"], 10000, 1, "tmin")
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)
    good = np.isnan(stn.stns["bad"])
    s = stn.stns[good]
    lst = np.column_stack([s["lst%02d" % m] for m in range(1, 13)])
    pts = ctx.make_pts(s["longitude"], s["latitude"], s["elevation"], s["tdi"], lst)
    j = np.arange(s.size, dtype=np.int32)
    t0 = time.time()
    _, norms, se, st = ctx.interp_points(_lib.TMIN, pts, excl=j, rm_zero_dist=True, daily=False)
    t1 = time.time()
    ok = st == 0
    obs_norm = np.column_stack([s["norm%02d" % m] for m in range(1, 13)])
    mae = np.abs(norms[ok] - obs_norm[ok]).mean()
    print("LOO xval normals, %d stations x 12 months: %.3f s host call (%.3g station-months/s), ok %d, MAE %.3f degC, status %s"
          % (s.size, t1 - t0, s.size * 12 / (t1 - t0), ok.sum(), mae, dict(zip(*np.unique(st, return_counts=True)))))
    db = orc.Db(stn)
    for q in (0, 1234, 9000):
        pt = orc.make_pt(s["longitude"][q], s["latitude"][q], s["elevation"][q], s["tdi"][q], lst[q])
        rc, _, n_o, se_o = orc.interp(db, orc.params(), pt, excl=int(q), rm_zero_dist=True, daily=False)
        assert rc == st[q]
        if rc == 0:
            assert np.abs(n_o - norms[q]).max() < 1e-4 and np.abs(se_o - se[q]).max() < 1e-4
    ctx.close()


if __name__ == "__main__":
    main()
