import sys, json, numpy as np
import torch
sys.path.insert(0, "/root/repo")
from topowx_amd import xval
res, arr = xval.run_config5(12000, 1, "tmin", 3000, 0, 1, 0, "cpu", db="c5")
nug, psill, rng = arr["vario"]
ok = np.isfinite(nug) & (psill > 0) & (rng > 0)
r = nug[ok] / psill[ok]
print("fitted: n", ok.sum(), "pure nugget frac", float(((psill == 0) | (rng == 0))[np.isfinite(nug)].mean()))
print("nug/psill quantiles", np.quantile(r, [0.01, 0.05, 0.25, 0.5, 0.75, 0.95]))
print("frac with 16 nug < psill (necessary condition):", float((16 * nug[ok] < psill[ok]).mean()))
print("range quantiles km", np.quantile(rng[ok], [0.05, 0.5, 0.95]), "nug q", np.quantile(nug[ok], [0.05, .5, .95]), "psill q", np.quantile(psill[ok], [.05, .5, .95]))
