"""sha256 of the outputs of the headline tile (C2: every kriging bucket has systems), default build and TWX_FLAG_UK_F64_ALL,
plus a point batch with explicit bandwidths on every kernel-size boundary: two builds of the library whose kernels
perform the same arithmetic in the same order print the same lines (tests/tools/ab_bits.sh).
    python3 tests/tools/gpu_out_hash.py"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from topowx_amd import _lib, synth  # noqa: E402


def h(*arrs):
    m = hashlib.sha256()
    for a in arrs:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:16]


grid, tmin, tmax = synth.make_case("C2")
for flags, name in ((0, "default"), (_lib.FLAG_UK_F64_ALL, "f64_all"), (_lib.FLAG_NO_HOST_SYNC, "no_host_sync")):
    ctx = _lib.Context(flags=flags)
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    out = ctx.interp_grid(grid, daily=False)
    print(name, "grid", h(out["norm_tmin"], out["se_tmin"], out["norm_tmax"], out["se_tmax"], out["status"]), int((out["status"] == 0).sum()))
    if flags == 0:
        cells = np.argwhere(np.asarray(grid["mask"]) != 0)[::97][:120]
        ks = np.array([7, 16, 33, 40, 41, 48, 49, 56, 57, 64, 65, 72, 73, 80, 81, 88, 89, 96, 97, 104, 105, 112, 113, 120, 121, 128, 129, 136, 137,
                       144, 145, 147, 150, 152] * 4, np.int32)[:len(cells)]
        pts = ctx.make_pts(grid["lon"][cells[:, 1]], grid["lat"][cells[:, 0]], grid["elev"][cells[:, 0], cells[:, 1]],
                           grid["tdi"][cells[:, 0], cells[:, 1]], grid["lst_night"][:, cells[:, 0], cells[:, 1]].T)
        for vario in (None, (0.05, 2.0, 900.0), (1e-3, 1.0, 40.0)):
            mean, var, used, st, _ = ctx.krig_points(_lib.TMIN, pts, 3, nnghs=ks, vario=None if vario is None else [vario] * len(ks))
            print(name, "points", vario, h(mean, var, used, st))
    ctx.close()
