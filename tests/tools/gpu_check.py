#!/usr/bin/env python3
"""Quick GPU-vs-oracle diagnostics (prints errors instead of asserting)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib  # noqa: E402
import make_golden  # noqa: E402


def step(name):
    def deco(f):
        t = time.time()
        try:
            f()
            print("[ok  ] %-28s %.2fs" % (name, time.time() - t), flush=True)
        except Exception:
            print("[FAIL] %s" % name)
            traceback.print_exc()
        return f
    return deco


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    grid, tmin, tmax = make_golden.case_inputs()
    orc.build()
    dbn, dbx = orc.Db(tmin), orc.Db(tmax)
    prm = orc.params()
    ctx = _lib.Context()
    print(_lib.load().twx_version().decode())
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)

    @step("knn vs golden")
    def _():
        bad = 0
        for k in (35, 100, 147):
            for rmz in (0, 1):
                sel = np.nonzero((g["sel_k"] == k) & (g["sel_rmz"] == rmz))[0]
                idx, dist, wgt, st = ctx.knn(_lib.TMIN, g["sel_lon"][sel], g["sel_lat"][sel], k,
                                             excl=g["sel_excl"][sel], rm_zero_dist=bool(rmz))
                bad += int((idx != g["sel_idx"][sel][:, :k]).sum()) + int((st != 0).sum())
                print("   k=%d rmz=%d idx mismatches %d  max|ddist| %.2e  max|dwgt| %.2e" % (
                    k, rmz, (idx != g["sel_idx"][sel][:, :k]).sum(), np.abs(dist - g["sel_dist"][sel][:, :k]).max(),
                    np.abs(wgt - g["sel_wgt"][sel][:, :k]).max()))
        assert bad == 0

    @step("krig_points vs golden")
    def _():
        cells, mth = g["kr_cell"], g["kr_mth"]
        pts = ctx.make_pts(grid["lon"][cells[:, 1]], grid["lat"][cells[:, 0]], grid["elev"][cells[:, 0], cells[:, 1]],
                           grid["tdi"][cells[:, 0], cells[:, 1]], grid["lst_night"][:, cells[:, 0], cells[:, 1]].T)
        mean, var, used, st, ngh = ctx.krig_points(_lib.TMIN, pts, mth, want_idx=True)
        print("   status", np.unique(st), "nnghs mismatches", (used != g["kr_nnghs"]).sum())
        print("   max|dmean| %.3e  max|dvar| %.3e" % (np.abs(mean - g["kr_mean"]).max(), np.abs(var - g["kr_var"]).max()))

    @step("grid normals vs oracle (24x24)")
    def _():
        rs, cs = slice(10, 34), slice(40, 64)
        t0 = time.time()
        want = orc.interp_grid(dbn, dbx, prm, grid, daily=False, nthreads=8, rows=rs, cols=cs)
        t1 = time.time()
        got = ctx.interp_grid(grid, daily=False, rows=rs, cols=cs)
        t2 = time.time()
        print("   oracle %.2fs  gpu(host api) %.2fs  timing %s" % (t1 - t0, t2 - t1, ctx.timing()))
        print("   status equal:", np.array_equal(want["status"], got["status"]), np.unique(got["status"]))
        for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
            print("   %-10s max|d| %.3e" % (k, np.abs(want[k].astype(np.float64) - got[k]).max()))

    @step("gwr/interp points vs golden")
    def _():
        cells = g["it_cell"]
        pts = ctx.make_pts(grid["lon"][cells[:, 1]], grid["lat"][cells[:, 0]], grid["elev"][cells[:, 0], cells[:, 1]],
                           grid["tdi"][cells[:, 0], cells[:, 1]], grid["lst_night"][:, cells[:, 0], cells[:, 1]].T)
        d, norms, se, st = ctx.interp_points(_lib.TMIN, pts)
        print("   status", st, "max|dnorm| %.3e max|dse| %.3e max|ddaily| %.3e" % (
            np.abs(norms - g["it_norms"]).max(), np.abs(se - g["it_se"]).max(), np.abs(d - g["it_daily"]).max()))
        c = dbn.cols
        j = g["xv_idx"]
        pts = ctx.make_pts(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j].T)
        d, norms, se, st = ctx.interp_points(_lib.TMIN, pts, excl=j, rm_zero_dist=True)
        print("   xval status", st, "max|dnorm| %.3e max|ddaily| %.3e" % (
            np.abs(norms - g["xv_norms"]).max(), np.abs(d - g["xv_daily"]).max()))

    @step("grid daily + fixer vs oracle (8x8)")
    def _():
        lo = make_golden.lowered_tmax(tmax)
        ctx2 = _lib.Context()
        ctx2.set_stations(_lib.TMIN, tmin)
        ctx2.set_stations(_lib.TMAX, lo)
        dbl = orc.Db(lo)
        rs, cs = slice(50, 58), slice(20, 28)
        want = orc.interp_grid(dbn, dbl, prm, grid, daily=True, nthreads=8, rows=rs, cols=cs)
        got = ctx2.interp_grid(grid, daily=True, rows=rs, cols=cs)
        print("   timing", ctx2.timing())
        print("   status equal", np.array_equal(want["status"], got["status"]), "ninvalid equal",
              np.array_equal(want["ninvalid"], got["ninvalid"]), want["ninvalid"].ravel()[:8], got["ninvalid"].ravel()[:8])
        for k in ("norm_tmin", "norm_tmax", "se_tmin"):
            print("   %-10s max|d| %.3e" % (k, np.abs(want[k].astype(np.float64) - got[k]).max()))
        for k in ("daily_tmin", "daily_tmax"):
            dd = np.abs(want[k].astype(int) - got[k].astype(int))
            print("   %-10s max LSB diff %d, frac equal %.6f" % (k, dd.max(), (dd == 0).mean()))
        ctx2.close()

    @step("fix_pair / pack vs golden")
    def _():
        days = tmin.days
        nd = days.size
        a = np.tile(g["fx_in_min"], (1, 3))[:, :nd]
        b = np.tile(g["fx_in_max"], (1, 3))[:, :nd]
        fa, fb, ninv, nmin, nmax, st = ctx.fix_pair(a, b)
        for i in range(a.shape[0]):
            rc, oa, ob, on = orc.fixer(a[i], b[i])
            print("   series %d ninv %d/%d max|d| %.2e status %d/%d" % (i, ninv[i], on, max(np.abs(fa[i] - oa).max(), np.abs(fb[i] - ob).max()), st[i], rc))
            if on > 0:
                wn = orc.recompute_norms(oa, dbn.day_month, dbn.day_year)
                print("      norms max|d| %.2e" % np.abs(wn - nmin[i]).max())
        print("   pack equal:", np.array_equal(ctx.pack_i16(g["pk_in"]), g["pk_out"]))

    ctx.close()


if __name__ == "__main__":
    main()
