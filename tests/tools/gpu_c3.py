#!/usr/bin/env python3
"""BASELINE.json configs[2] on ONE GPU: full CONUS-shaped 3250x7000 30-arcsec grid (~13 M valid
cells), 12 000 stations per variable, 12 monthly Tmin + Tmax normals.  Prints the device time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from topowx_amd import _lib, synth  # noqa: E402


def main():
    t0 = time.time()
    grid = synth.make_grid("C3")
    tmin = synth.make_stations(grid["bbox"], 12000, 2, "tmin")
    tmax = synth.make_stations(grid["bbox"], 12000, 2, "tmax")
    nvalid = int((grid["mask"] != 0).sum())
    print("synthetic C3 in %.0f s: %d x %d cells, %d valid (%.0f %%)" % (
        time.time() - t0, grid["mask"].shape[0], grid["mask"].shape[1], nvalid, 100.0 * nvalid / grid["mask"].size), flush=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    t0 = time.time()
    out = ctx.interp_grid(grid)
    t1 = time.time()
    tm = ctx.timing()
    st = out["status"]
    print("host call %.1f s (incl. H2D/D2H of %.1f GB), device %.2f s: uk %.2f select %.2f tile %.2f"
          % (t1 - t0, (grid["mask"].size * (105 + 196)) / 1e9, tm["total_ms"] / 1e3, tm["uk_ms"] / 1e3,
             tm["select_ms"] / 1e3, tm["tile_cand_ms"] / 1e3))
    print("status:", dict(zip(*np.unique(st, return_counts=True))))
    print("C3 normals: %.3g cell-months/s on one GPU (device), %d kriging systems" % (
        nvalid * 24 / (tm["total_ms"] * 1e-3), tm["uk_solves"]))
    ok = st == 0
    print("norm_tmin Jan range %.2f..%.2f, Jul %.2f..%.2f; se mean %.3f" % (
        out["norm_tmin"][0][ok].min(), out["norm_tmin"][0][ok].max(), out["norm_tmin"][6][ok].min(),
        out["norm_tmin"][6][ok].max(), out["se_tmin"][0][ok].mean()))
    ctx.close()


if __name__ == "__main__":
    main()
