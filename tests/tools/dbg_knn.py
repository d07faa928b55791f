import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from topowx_amd import _lib
import make_golden
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
grid, tmin, tmax = make_golden.case_inputs()
ctx = _lib.Context(); ctx.set_stations(_lib.TMIN, tmin)
k = 35
sel = np.nonzero((g["sel_k"] == k) & (g["sel_rmz"] == 1))[0]
idx, dist, wgt, st = ctx.knn(_lib.TMIN, g["sel_lon"][sel], g["sel_lat"][sel], k, rm_zero_dist=True)
idx0, dist0, _, _ = ctx.knn(_lib.TMIN, g["sel_lon"][sel], g["sel_lat"][sel], k, rm_zero_dist=False)
for i, s in enumerate(sel):
    w = g["sel_idx"][s][:k]
    print(i, "mism", (idx[i] != w).sum(), "st", st[i], "mindist gpu %.4f gold %.4f rmz0 %.4f" % (dist[i].min(), g["sel_dist"][s][:k].min(), dist0[i].min()),
          "maxdist gpu %.3f gold %.3f" % (dist[i].max(), g["sel_dist"][s][:k].max()))
    if (idx[i] != w).any():
        print("   gpu ", np.sort(idx[i])[:12], "\n   gold", np.sort(w)[:12], "\n   only gpu", np.setdiff1d(idx[i], w), "only gold", np.setdiff1d(w, idx[i]))
