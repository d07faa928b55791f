"""Error of twx_krig_points against the fp64 oracle on the close-pair table of tests/test_gpu_closepairs.py, per
(k, nugget, psill, range), next to the amplification psill / (2 (nug + psill (1 - exp(-hmin / range)))) that the
library's precise-build criterion (twx_uk.h: uk_needs_f64) is calibrated on.  Writes gpurun_out/closepair_scan.json.

    python tests/tools/gpu_closepair_scan.py [--fast-only]     (on the GPU box)
"""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: see tests/conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import make_golden
    from oracle import pyoracle as orc
    from test_gpu_closepairs import closepair_db
    from topowx_amd import _lib
    orc.build()
    grid, tmin, _ = make_golden.case_inputs()
    cells = np.array([(20, 30), (50, 70), (80, 15)])
    db = closepair_db(tmin, grid, cells)
    flags = 0
    if "--fast-only" in sys.argv:
        flags = getattr(_lib, "FLAG_UK_FAST_ONLY", 0)
    ctx = _lib.Context(flags=flags) if flags else _lib.Context()
    ctx.set_stations(_lib.TMIN, db, with_obs=False)
    odb, prm = orc.Db(db), orc.params()
    r, c = cells[:, 0], cells[:, 1]
    base = ctx.make_pts(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c].T)
    ks = (40, 72, 112, 147)
    pts = np.repeat(base, len(ks))
    kk = np.tile(np.array(ks, np.int32), len(cells))
    rows = []
    for nug in (0.0, 1e-4, 1e-3, 1e-2, 0.05, 0.2):
        for ps in (0.2, 2.0):
            for rg in (5.0, 40.0, 200.0, 900.0):
                vario = (nug, ps, rg)
                mean, var, used, st, _ = ctx.krig_points(_lib.TMIN, pts, 3, nnghs=kk, vario=[vario] * pts.size)
                errs = []
                for i in range(pts.size):
                    rr, cc = cells[i // len(ks)]
                    pt = orc.make_pt(grid["lon"][cc], grid["lat"][rr], grid["elev"][rr, cc], grid["tdi"][rr, cc],
                                     grid["lst_night"][:, rr, cc])
                    rc, m, v, u, _ = orc.krig(odb, prm, pt, 3, nnghs=int(kk[i]), vario=vario)
                    if rc or st[i]:
                        errs.append(float("nan"))
                    else:
                        errs.append(max(abs(mean[i] - m), abs(var[i] - v)))
                amp = ps / (2 * (nug + ps * -np.expm1(-0.05 / rg)))
                rows.append(dict(nug=nug, psill=ps, rng=rg, amp_hmin50m=amp, err_max=float(np.nanmax(errs)),
                                 err_by_k={str(k): float(np.nanmax(errs[j::len(ks)])) for j, k in enumerate(ks)}))
                print("nug %-7g psill %-4g rng %-5g amp %9.3g  err %.3g   %s" % (
                    nug, ps, rg, amp, rows[-1]["err_max"], " ".join("%.2g" % rows[-1]["err_by_k"][str(k)] for k in ks)), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = "closepair_scan_fast.json" if flags else "closepair_scan.json"
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
