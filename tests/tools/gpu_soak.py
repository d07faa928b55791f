"""Soak: seeded random small cases through the whole grid path (normals + daily + fixer), GPU against the CPU oracle.
Every case draws its own grid extent and shape (not multiples of the 8-cell tiles), mask, station count, nugget scale
(down to values that route systems to the fp64 covariance build), close station pairs, Tmax offset (days with
tmin >= tmax for the fixer), batch size and flags; some cases have too few stations or NaN predictors (failure statuses).
No cell is set aside: the fixer's test tmin >= tmax is discontinuous, but since round 6 the library re-kriges every cell with
a day of |tmax - tmin| < 2e-5 degC on the fp64 covariance build (the tie guard, include/twx.h), so ninvalid must be the oracle's in
every cell; a case whose ninvalid differs FAILS, and the margins of the differing cells are reported.  Not part of the test suite (minutes of oracle time): run on the GPU
box after kernel changes.   python3 tests/tools/gpu_soak.py [n_cases] [first_seed]  ->  gpurun_out/soak.json"""
import datetime as dt
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib, stationdb as sdb, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
orc.build()
nthr = min(64, os.cpu_count() or 8)
report, worst = [], {"norm": 0.0, "se": 0.0, "flip": 0.0}
bad = 0
for seed in range(seed0, seed0 + ncase):
    rng = np.random.default_rng(9000 + seed)
    Y, X = int(rng.integers(20, 61)), int(rng.integers(20, 61))
    grid = synth.make_grid("C1", nrows=Y, ncols=X, lat_north=float(rng.uniform(31, 48)), lon_west=float(rng.uniform(-120, -80)),
                           seed=100 + seed)
    mask = grid["mask"].copy()
    for _ in range(int(rng.integers(0, 4))):                              # masked rectangles
        r, c = int(rng.integers(0, Y)), int(rng.integers(0, X))
        mask[r:r + int(rng.integers(1, 12)), c:c + int(rng.integers(1, 12))] = 0
    grid["mask"] = mask
    years = int(rng.integers(1, 3))
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1980 + years, 12, 31))
    n = int(rng.integers(220, 700))
    if rng.random() < 0.12:                                              # too few stations for some / all bandwidths: failure statuses
        n = int(rng.integers(12, 160))
    if rng.random() < 0.15:                                              # NaN predictors in patches of the grid
        for name in ("elev", "tdi"):
            if rng.random() < 0.5:
                r, c = int(rng.integers(0, Y)), int(rng.integers(0, X))
                grid[name] = grid[name].copy()
                grid[name][r:r + int(rng.integers(1, 6)), c:c + int(rng.integers(1, 6))] = np.nan
        if rng.random() < 0.5:
            r, c = int(rng.integers(0, Y)), int(rng.integers(0, X))
            grid["lst_day"] = grid["lst_day"].copy()
            grid["lst_day"][int(rng.integers(0, 12)), r:r + 3, c:c + 3] = np.nan
    tmin = synth.make_stations(grid["bbox"], n, 300 + seed, "tmin", days, with_obs=True, expand_deg=float(rng.uniform(0.3, 1.5)))
    tmax = synth.make_stations(grid["bbox"], n, 300 + seed, "tmax", days, with_obs=True, expand_deg=float(rng.uniform(0.3, 1.5)))
    knobs = {"seed": seed, "Y": Y, "X": X, "stations": n, "years": years}
    dbs = []
    for db, var in ((tmin, "tmin"), (tmax, "tmax")):
        stns, obs = db.stns.copy(), db.var.copy()
        if rng.random() < 0.6:                                            # small nuggets: ill-conditioned systems
            scale = 10.0 ** rng.uniform(-4, 0)
            for m in range(1, 13):
                stns[sdb.get_krigparam_varname(m, sdb.VARIO_NUG)] *= scale
            knobs["nug_scale_" + var] = scale
        if rng.random() < 0.5:                                            # station pairs 50-300 m apart
            k = int(rng.integers(2, 8))
            src = rng.choice(stns.size, k, replace=False)
            twin = stns[src].copy()
            twin[sdb.LON] += rng.uniform(5e-4, 3e-3, k)
            twin[sdb.STN_ID] = ["Z%07d" % i for i in range(k)]
            stns = np.concatenate([stns, twin])
            obs = np.concatenate([obs, obs[:, src] + rng.normal(0, 0.2, (obs.shape[0], k)).astype(np.float32)], axis=1)
            knobs["twins_" + var] = k
        if var == "tmax" and rng.random() < 0.6:                          # inverted days for the fixer
            shift = float(rng.uniform(4, 9))
            for m in range(1, 13):
                stns[sdb.get_norm_varname(m)] -= shift
            obs = obs - np.float32(shift)
            knobs["tmax_shift"] = shift
        dbs.append(sdb.StationDataWrkChk(stns, var, days, obs))
    flags = int(rng.choice([0, 0, _lib.FLAG_NO_HOST_SYNC, _lib.FLAG_DAILY_GATHER, _lib.FLAG_OBS_ADDR64, _lib.FLAG_UK_F64_ALL,
                            _lib.FLAG_UK_F64_ALL | _lib.FLAG_NO_HOST_SYNC]))
    batch = int(rng.choice([0, 0, 512, 2048]))
    tile_cells = int(rng.choice([0, 0, 0, 4, 16]))                       # 8 (tile tables), 4 (small tables), 16 (no tables: strips)
    which = [("tmin", "tmax"), ("tmin", "tmax"), ("tmin", "tmax"), ("tmin",), ("tmax",)][int(rng.integers(0, 5))]
    knobs.update(flags=flags, batch_cells=batch, tile_cells=tile_cells, variables="+".join(which))
    t0 = time.perf_counter()
    ctx = _lib.Context(flags=flags, batch_cells=batch, tile_cells=tile_cells)
    ctx.set_stations(_lib.TMIN, dbs[0])
    ctx.set_stations(_lib.TMAX, dbs[1])
    got = ctx.interp_grid(grid, variables=which, daily=True)
    tim = ctx.timing()
    ctx.close()
    t1 = time.perf_counter()
    want = orc.interp_grid(orc.Db(dbs[0]) if "tmin" in which else None, orc.Db(dbs[1]) if "tmax" in which else None, orc.params(),
                           grid, daily=True, nthreads=nthr)
    t2 = time.perf_counter()
    rec = dict(knobs, gpu_s=round(t1 - t0, 2), oracle_s=round(t2 - t1, 2))
    ok = got["status"] == 0
    rec["status_equal"] = bool(np.array_equal(got["status"], want["status"]))
    if not rec["status_equal"]:                                           # which codes disagree, where
        neq = got["status"] != want["status"]
        pairs, cnt = np.unique(np.stack([got["status"][neq], want["status"][neq]]), axis=1, return_counts=True)
        rec["status_mismatch_gpu_vs_oracle"] = [[int(a), int(b), int(c)] for (a, b), c in zip(pairs.T, cnt)]
        rr, cc = np.nonzero(neq)
        rec["status_mismatch_first_cell"] = [int(rr[0]), int(cc[0])]
    rec["cells_ok"] = int(ok.sum())
    rec["failed_cells"] = int((got["status"] > 0).sum())
    rec["ninvalid_equal"] = bool(np.array_equal(got["ninvalid"], want["ninvalid"]))
    rec["ninvalid_max"] = int(want["ninvalid"][want["status"] == 0].max()) if (want["status"] == 0).any() else 0
    if not rec["ninvalid_equal"] and len(which) == 2:
        # the fixer's test is tmin >= tmax on two interpolated fp64 series: how close to a tie is the nearest day of the cells
        # that disagree?  (a tie broken the other way moves that day and its 15-day tails: a discontinuity of the algorithm,
        # not an error of either side)
        rr, cc = np.nonzero((got["ninvalid"] != want["ninvalid"]) & ok)
        margins = []
        odn, odx, prm = orc.Db(dbs[0]), orc.Db(dbs[1]), orc.params()
        for r, c in list(zip(rr, cc))[:64]:
            ptn = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c])
            ptx = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_day"][:, r, c])
            _, dn, _, _ = orc.interp(odn, prm, ptn)
            _, dx, _, _ = orc.interp(odx, prm, ptx)
            gap = dx - dn
            margins.append({"cell": [int(r), int(c)], "ninvalid_gpu": int(got["ninvalid"][r, c]), "ninvalid_oracle": int(want["ninvalid"][r, c]),
                            "oracle_days_tmin_ge_tmax": int((gap <= 0).sum()), "smallest_abs_gap_degC": float(np.abs(gap).min())})
        rec["ninvalid_mismatch_cells"] = int(rr.size)
        rec["ninvalid_mismatch_margins"] = margins
        rec["near_tie_cells"] = int(sum(1 for m in margins if m["smallest_abs_gap_degC"] < 2e-5))      # (diagnostic only)
    rec["f64_solves"] = int(tim.get("uk_f64_solves", -1)) if isinstance(tim, dict) else int(getattr(tim, "uk_f64_solves", -1))
    rec["tie_cells"] = int(tim.get("tie_cells", -1)) if isinstance(tim, dict) else -1
    for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
        d = np.abs(got[k].astype(np.float64) - want[k])[:, ok] if k in got else np.zeros(0)
        rec[k] = float(d.max()) if d.size else 0.0
    for k in ("daily_tmin", "daily_tmax"):
        d = np.abs(got[k].astype(np.int64) - want[k].astype(np.int64))[:, ok] if k in got else np.zeros(0)
        rec[k + "_maxdiff"] = int(d.max()) if d.size else 0
        rec[k + "_flip"] = float((d != 0).mean()) if d.size else 0.0
    good = (rec["status_equal"] and rec["ninvalid_equal"] and max(rec["norm_tmin"], rec["norm_tmax"], rec["se_tmin"], rec["se_tmax"]) < 1e-4
            and max(rec["daily_tmin_maxdiff"], rec["daily_tmax_maxdiff"]) <= 1 and max(rec["daily_tmin_flip"], rec["daily_tmax_flip"]) < 1e-3)
    if flags & _lib.FLAG_UK_F64_ALL:                                      # fp64 build everywhere: the oracle's f4 / int16 bits
        exact = all(np.array_equal(got[k][:, ok], want[k].astype(np.float32)[:, ok]) for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax") if k in got)
        flips = sum(int((got[k] != want[k])[:, ok].sum()) for k in ("daily_tmin", "daily_tmax") if k in got)
        rec["f64_all_exact_f4"] = bool(exact)
        rec["f64_all_int16_flips"] = flips
    rec["pass"] = bool(good)
    bad += not good
    worst["norm"] = max(worst["norm"], rec["norm_tmin"], rec["norm_tmax"])
    worst["se"] = max(worst["se"], rec["se_tmin"], rec["se_tmax"])
    worst["flip"] = max(worst["flip"], rec["daily_tmin_flip"], rec["daily_tmax_flip"])
    report.append(rec)
    print(json.dumps(rec), flush=True)
out = {"cases": len(report), "failed": bad, "worst_abs_degC_norm": worst["norm"], "worst_abs_degC_se": worst["se"],
       "worst_int16_flip_rate": worst["flip"], "records": report}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "soak.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "records"}))
sys.exit(1 if bad else 0)
