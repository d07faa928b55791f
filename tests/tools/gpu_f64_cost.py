"""Cost of the fp64 covariance build: the C2 tile with a station table whose variograms flag EVERY system
(nugget 1e-3, psill 1, range 400 km) against the same table on the fast build (TWX_FLAG_UK_FAST_ONLY).
    python tests/tools/gpu_f64_cost.py        (on the GPU box)"""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from topowx_amd import _lib, stationdb as sdb, synth
    grid = synth.make_grid("C2")
    stn = synth.make_stations(grid["bbox"], 10000, 1, "tmin")
    for m in range(1, 13):
        ok = np.isfinite(stn.stns[sdb.get_krigparam_varname(m, sdb.VARIO_NUG)])
        stn.stns[sdb.get_krigparam_varname(m, sdb.VARIO_NUG)][ok] = 1e-3
        stn.stns[sdb.get_krigparam_varname(m, sdb.VARIO_PSILL)][ok] = 1.0
        stn.stns[sdb.get_krigparam_varname(m, sdb.VARIO_RNG)][ok] = 400.0
    out = {}
    res = {}
    for name, flags in (("fp64_build", 0), ("fast_only", _lib.FLAG_UK_FAST_ONLY)):
        ctx = _lib.Context(flags=flags)
        ctx.set_stations(_lib.TMIN, stn, with_obs=False)
        for _ in range(3):
            got = ctx.interp_grid(grid, variables=("tmin",), daily=False)
            t = ctx.timing()
        res[name] = got
        out[name] = {"uk_ms": t["uk_ms"], "uk_solves": t["uk_solves"], "uk_f64_solves": t["uk_f64_solves"],
                     "cells_ok": int((got["status"] == 0).sum())}
        ctx.close()
    ok = (res["fp64_build"]["status"] == 0) & (res["fast_only"]["status"] == 0)
    d = np.abs(res["fp64_build"]["norm_tmin"].astype(np.float64) - res["fast_only"]["norm_tmin"])[:, ok]
    out["fast_vs_fp64_max_abs_degC"] = float(d.max())
    out["ratio"] = out["fp64_build"]["uk_ms"] / out["fast_only"]["uk_ms"]
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f64_cost.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
