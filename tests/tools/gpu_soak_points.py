"""Soak of the point entries (twx_krig_points, twx_gwr_points, twx_interp_points, twx_fit_vario_points), GPU against the CPU
oracle on seeded random cases: random station databases (some too small, some with tiny nuggets and twin stations), points on
stations (with and without leave-one-out / zero-distance removal) and off them, NaN predictors, random months, automatic
and explicit bandwidths (also out of range), given and smoothed variograms.  Not part of the suite.
python3 tests/tools/gpu_soak_points.py [n_cases] [first_seed]  ->  gpurun_out/soak_points.json"""
import datetime as dt
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import _lib, stationdb as sdb, synth  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
orc.build()
TOL = 1e-4
report, bad = [], 0
tot = {"krig": 0, "gwr": 0, "interp": 0, "vario": 0, "failed_status": 0}
for seed in range(seed0, seed0 + ncase):
    rng = np.random.default_rng(40000 + seed)
    lat0, lon0 = float(rng.uniform(31, 47)), float(rng.uniform(-120, -80))
    bbox = (lat0, lat0 + 0.6, lon0, lon0 + 0.8)
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1981, 12, 31))
    n = int(rng.integers(230, 600)) if rng.random() > 0.15 else int(rng.integers(20, 170))
    db = synth.make_stations(bbox, n, 500 + seed, "tmin", days, with_obs=True, expand_deg=float(rng.uniform(0.4, 1.2)))
    stns, obs = db.stns.copy(), db.var.copy()
    knobs = {"seed": seed, "stations": n}
    if rng.random() < 0.5:
        scale = 10.0 ** rng.uniform(-4, 0)
        for m in range(1, 13):
            stns[sdb.get_krigparam_varname(m, sdb.VARIO_NUG)] *= scale
        knobs["nug_scale"] = scale
    if rng.random() < 0.5:
        k = int(rng.integers(2, 8))
        src = rng.choice(stns.size, k, replace=False)
        twin = stns[src].copy()
        twin[sdb.LON] += rng.uniform(5e-4, 3e-3, k)
        twin[sdb.STN_ID] = ["Z%07d" % i for i in range(k)]
        stns = np.concatenate([stns, twin])
        obs = np.concatenate([obs, obs[:, src] + rng.normal(0, 0.2, (obs.shape[0], k)).astype(np.float32)], axis=1)
        knobs["twins"] = k
    db = sdb.StationDataWrkChk(stns, "tmin", days, obs)
    odb, prm = orc.Db(db), orc.params()
    c = odb.cols
    ns = c["lon"].size
    # points: stations (leave-one-out or not) and random locations
    npt = 48
    on = rng.random(npt) < 0.5
    j = rng.integers(0, ns, npt)
    lon = np.where(on, c["lon"][j], rng.uniform(bbox[2], bbox[3], npt))
    lat = np.where(on, c["lat"][j], rng.uniform(bbox[0], bbox[1], npt))
    elev = np.where(on, c["elev"][j], synth.field_elev(lon, lat))
    tdi = np.where(on, c["tdi"][j], synth.field_tdi(lon, lat))
    lst = np.where(on[:, None], c["lst"][:, j].T, c["lst"][:, rng.integers(0, ns, npt)].T + rng.normal(0, 0.5, (npt, 12)))
    nanp = rng.random(npt) < 0.06
    elev = np.where(nanp, np.nan, elev)
    excl = np.where(on & (rng.random(npt) < 0.7), j, -1).astype(np.int32)
    rmz = bool(rng.random() < 0.5)
    mth = rng.integers(1, 13, npt).astype(np.int32)
    nn = np.where(rng.random(npt) < 0.5, 0, rng.integers(8, 175, npt)).astype(np.int32)
    over = nn > 152                                          # library limit (TWX_MAX_NNGHS; the reference ladder ends at 147): TWX_CELL_RANGE
    given = rng.random(npt) < 0.3
    vario = np.full((npt, 3), np.nan)
    vario[given] = np.column_stack([10.0 ** rng.uniform(-4, 0, given.sum()), rng.uniform(0.2, 2.0, given.sum()),
                                    np.where(rng.random(given.sum()) < 0.1, 0.0, rng.uniform(5, 900, given.sum()))])
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, db)
    pts = ctx.make_pts(lon, lat, elev, tdi, lst)
    opt = [orc.make_pt(lon[i], lat[i], elev[i], tdi[i], lst[i]) for i in range(npt)]
    rec = dict(knobs, rm_zero=rmz, worst={})
    ok_case = True

    def note(name, cond, detail):
        global ok_case
        if not cond:
            ok_case = False
            rec.setdefault("mismatch", []).append([name, detail])

    # ---- kriging
    mean, var, used, st, _ = ctx.krig_points(_lib.TMIN, pts, mth, nnghs=nn, vario=vario, excl=excl, rm_zero_dist=rmz)
    w = 0.0
    for i in range(npt):
        rc, m_, v_, k_, _ = orc.krig(odb, prm, opt[i], int(mth[i]), int(nn[i]), None if not given[i] else list(vario[i]),
                                     int(excl[i]), rmz)
        if over[i]:
            note("krig_range", st[i] == 6, [i, int(st[i])])
            continue
        note("krig_status", st[i] == rc, [i, int(st[i]), int(rc), int(nn[i]), bool(given[i]), bool(nanp[i])])
        if rc == 0 and st[i] == 0:
            note("krig_used", used[i] == k_, [i, int(used[i]), int(k_)])
            w = max(w, abs(mean[i] - m_), abs(var[i] - v_))
        tot["failed_status"] += rc != 0
    note("krig_value", w < TOL, w)
    rec["worst"]["krig"] = w
    tot["krig"] += npt
    # ---- GWR anomalies of one month
    pn = rng.normal(5, 8, npt)
    out, used, st = ctx.gwr_points(_lib.TMIN, pts, pn, mth, nnghs=nn, excl=excl, rm_zero_dist=rmz)
    w = 0.0
    for i in range(npt):
        rc, o_, k_, _, _ = orc.gwr_mth(odb, prm, opt[i], float(pn[i]), int(mth[i]), int(nn[i]), int(excl[i]), rmz)
        if over[i]:
            note("gwr_range", st[i] == 6, [i, int(st[i])])
            continue
        note("gwr_status", st[i] == rc, [i, int(st[i]), int(rc), int(nn[i]), bool(nanp[i])])
        if rc == 0 and st[i] == 0:
            note("gwr_used", used[i] == k_, [i, int(used[i]), int(k_)])
            w = max(w, float(np.abs(out[i, :o_.size] - o_).max()))
    note("gwr_value", w < TOL, w)
    rec["worst"]["gwr"] = w
    tot["gwr"] += npt
    # ---- the whole point interpolation (normals + SE + daily), a dozen points
    sub = slice(0, 12)
    d, norms, se, st = ctx.interp_points(_lib.TMIN, pts[sub], excl=excl[sub], rm_zero_dist=rmz)
    w = 0.0
    for i in range(12):
        rc, d_, n_, s_ = orc.interp(odb, prm, opt[i], int(excl[i]), rmz)
        note("interp_status", st[i] == rc, [i, int(st[i]), int(rc), bool(nanp[i])])
        if rc == 0 and st[i] == 0:
            w = max(w, float(np.abs(norms[i] - n_).max()), float(np.abs(se[i] - s_).max()), float(np.abs(d[i] - d_).max()))
    note("interp_value", w < TOL, w)
    rec["worst"]["interp"] = w
    tot["interp"] += 12
    # ---- variogram fit at station records (step22 shape: automatic bandwidth, the station stays in)
    sj = rng.integers(0, ns, 12)
    spts = ctx.make_pts(c["lon"][sj], c["lat"][sj], c["elev"][sj], c["tdi"][sj], c["lst"][:, sj].T)
    smth = rng.integers(1, 13, 12).astype(np.int32)
    vf, _, st = ctx.fit_vario_points(_lib.TMIN, spts, smth)
    w = 0.0
    for i in range(12):
        rc, v_, _ = orc.build_krig_params(odb, prm, orc.make_pt(c["lon"][sj[i]], c["lat"][sj[i]], c["elev"][sj[i]], c["tdi"][sj[i]],
                                                               c["lst"][:, sj[i]]), int(smth[i]))
        note("vario_status", st[i] == rc, [i, int(st[i]), int(rc)])
        if rc == 0 and st[i] == 0:
            w = max(w, float(np.max(np.abs(vf[i] - v_) / (np.abs(v_) + 1e-9))))
    # (typically 1e-7; the Gauss-Newton iteration on the range amplifies the 1e-7 differences of the GLS residuals where the
    # weighted SSE is flat: a handful of fits in a thousand land 1e-6 ... 1e-4 from the oracle's, the kriged values stay put)
    note("vario_value", w < 1e-3, w)
    rec["vario_above_1e-6"] = bool(w > 1e-6)
    rec["worst"]["vario_rel"] = w
    tot["vario"] += 12
    # ---- step21 shape: leave-one-out, explicit bandwidth, fit then krige all twelve months (KrigTairAll.krigall)
    w = 0.0
    for sj1 in rng.integers(0, ns, 5):
        k = int(rng.choice([35, 42, 57, 76, 101, 147]))
        p12 = ctx.make_pts(np.repeat(c["lon"][sj1], 12), np.repeat(c["lat"][sj1], 12), np.repeat(c["elev"][sj1], 12),
                           np.repeat(c["tdi"][sj1], 12), np.tile(c["lst"][:, sj1], (12, 1)))
        m12 = np.arange(1, 13, dtype=np.int32)
        vf, _, st1 = ctx.fit_vario_points(_lib.TMIN, p12, m12, nnghs=k, excl=int(sj1), rm_zero_dist=True)
        mean, _, _, st2, _ = ctx.krig_points(_lib.TMIN, p12, m12, nnghs=k, vario=np.nan_to_num(vf), excl=int(sj1), rm_zero_dist=True)
        rc, norms_o, _ = orc.krigall(odb, prm, orc.make_pt(c["lon"][sj1], c["lat"][sj1], c["elev"][sj1], c["tdi"][sj1], c["lst"][:, sj1]),
                                     k, excl=int(sj1), rm_zero_dist=True)
        gpu_ok = bool(np.all(st1 == 0) and np.all(st2 == 0))
        note("krigall_status", gpu_ok == (rc == 0), [int(sj1), k, st1.tolist(), st2.tolist(), int(rc)])
        if gpu_ok and rc == 0:
            w = max(w, float(np.abs(mean - norms_o).max()))
    note("krigall_value", w < TOL, w)
    rec["worst"]["krigall"] = w
    tot["krigall"] = tot.get("krigall", 0) + 5
    ctx.close()
    rec["pass"] = ok_case
    bad += not ok_case
    report.append(rec)
    print(json.dumps(rec), flush=True)
out = {"cases": len(report), "failed": bad, "points": tot,
       "worst": {k: max(r["worst"][k] for r in report) for k in ("krig", "gwr", "interp", "vario_rel", "krigall")}, "records": report}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "soak_points.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "records"}))
sys.exit(1 if bad else 0)
