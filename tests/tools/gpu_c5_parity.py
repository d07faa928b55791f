"""Config 5 (BASELINE.json configs[4]) at its own size against the CPU oracle: step21 and step22 for EVERY cross-validated
station of the 12 000-station seed-2 database -- 11 751 stations x 16 bandwidths x 12 months = 2.26 M variogram fits + kriged
normals, the bandwidth choice per (climate division, month), 141 k station variograms, step23's 2.26 M leave-one-out GWR
series (MAE / bias) and the bandwidths THEY choose, step24's normals + SE + daily values at every station -- the suite
compares a handful of stations.  The oracle's share runs on the box's host threads (ctypes releases the GIL).
    python3 tests/tools/gpu_c5_parity.py [--nstns 12000] [--max-stations N]  ->  gpurun_out/c5_parity.json"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pyoracle as orc  # noqa: E402
from topowx_amd import stationdb as sdb, synth, xval  # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


nstns, nmax = arg("--nstns", 12000), arg("--max-stations", 0)
orc.build()
bbox, seed = xval.config5_bbox("c5")
import datetime as dt  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402
days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1983, 12, 31))
stn = synth.make_stations(bbox, nstns, seed, "tmin", days, with_obs=True)
ids = xval.xval_station_ids(stn)
if nmax:
    ids = ids[:nmax]
ladder = np.asarray(xval.DFLT_LADDER)
res = {"stations_in_db": int(stn.stns.size), "stations": int(len(ids)), "ladder": [int(k) for k in ladder]}
nthr = min(256, os.cpu_count() or 8)

# ---- step21 on the GPU ----------------------------------------------------------------------------------
t0 = time.perf_counter()
_, mae = xval.optim_nstns_norms(stn, "tmin", stn_ids=ids)                  # [12, nb, n]
res["step21_gpu_s"] = round(time.perf_counter() - t0, 2)
before = stn.stns.copy()

# ---- step21 by the oracle: krigall per (station, bandwidth), all twelve months per call --------------------------
good = np.isnan(before[sdb.BAD])
odb, prm = orc.Db(sdb.StationSerialDataDb(before, "tmin", stn.days, None)), orc.params()
c = odb.cols
idx = {s: i for i, s in enumerate(before[sdb.STN_ID][good])}
obs = c["norm"]                                                            # [12, n good]


def one(q):
    j = idx[ids[q]]
    pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
    out = np.full((12, ladder.size), np.nan)
    ok = True
    for x, k in enumerate(ladder):
        rc, nrm, _ = orc.krigall(odb, prm, pt, int(k), excl=j, rm_zero_dist=True)
        if rc:
            ok = False
        else:
            out[:, x] = np.abs(nrm - obs[:, j])
    if not ok:                                                             # (the farm drops a station that fails anywhere)
        out[:] = np.nan
    return out


t0 = time.perf_counter()
with ThreadPoolExecutor(nthr) as ex:
    want = np.stack(list(ex.map(one, range(len(ids)))), axis=2)            # [12, nb, n]
res["step21_oracle_s"] = round(time.perf_counter() - t0, 1)
res["oracle_threads"] = nthr
fin_g, fin_o = np.isfinite(mae), np.isfinite(want)
res["step21_failed_stations_gpu"] = int((~fin_g).all(axis=(0, 1)).sum())
res["step21_failed_stations_equal"] = bool(np.array_equal(fin_g, fin_o))
both = fin_g & fin_o
d = np.abs(mae - want)[both]
res["step21_values"] = int(both.sum())
res["step21_abs_error_max_abs_diff_degC"] = float(d.max())
res["step21_abs_error_p99.99_diff_degC"] = float(np.quantile(d, 0.9999))

# ---- the bandwidth choice per (climate division, month): an integer outcome ----------------------------------
sg, chosen_g = xval.set_optim_nstns(before.copy(), ids, mae, ladder, sdb.get_optim_varname)
so, chosen_o = xval.set_optim_nstns(before.copy(), ids, want, ladder, sdb.get_optim_varname)
ndiv = len(chosen_g)
neq = sum(int((chosen_g[k] != chosen_o[k]).sum()) for k in chosen_g)
res["optim_nnghs_divisions"] = ndiv
res["optim_nnghs_choices"] = ndiv * 12
res["optim_nnghs_choices_differing"] = neq
if neq:
    # a differing choice is a tie of two bandwidths' mean errors within the two sides' agreement: how close?
    gaps = []
    pos = {s: i for i, s in enumerate(before[sdb.STN_ID])}
    div = before[sdb.CLIMDIV][[pos[s] for s in ids]]
    for k in chosen_g:
        for m in np.nonzero(chosen_g[k] != chosen_o[k])[0]:
            sel = div == k
            mm = np.nanmean(want[m][:, sel], axis=1)
            a, b = np.nonzero(ladder == chosen_g[k][m])[0][0], np.nonzero(ladder == chosen_o[k][m])[0][0]
            gaps.append(float(abs(mm[a] - mm[b])))
    res["optim_nnghs_differing_mean_error_gaps_degC"] = sorted(gaps)[:20]

# ---- step22: every station's variogram with the optimised bandwidths (the GPU's choice on both sides) --------------
stn.stns = sg
sg_before = sg.copy()                                                      # (the farm writes the variogram columns in place)
t0 = time.perf_counter()
_, nug, psill, rng = xval.set_stn_variograms(stn, "tmin", stn_ids=ids)
res["step22_gpu_s"] = round(time.perf_counter() - t0, 2)
odb2 = orc.Db(sdb.StationSerialDataDb(sg_before, "tmin", stn.days, None))
c2 = odb2.cols


def fit(q):
    j = idx[ids[q]]
    pt = orc.make_pt(c2["lon"][j], c2["lat"][j], c2["elev"][j], c2["tdi"][j], c2["lst"][:, j])
    out = np.full((3, 12), np.nan)
    for m in range(1, 13):
        rc, v, _ = orc.build_krig_params(odb2, prm, pt, m)
        if rc:
            out[:] = np.nan
            break
        out[:, m - 1] = v
    return out


t0 = time.perf_counter()
with ThreadPoolExecutor(nthr) as ex:
    wv = np.stack(list(ex.map(fit, range(len(ids)))), axis=2)              # [3, 12, n]
res["step22_oracle_s"] = round(time.perf_counter() - t0, 1)
gv = np.stack([nug, psill, rng])
fg, fo = np.isfinite(gv), np.isfinite(wv)
res["step22_failed_equal"] = bool(np.array_equal(fg, fo))
both = fg & fo
rel = np.abs(gv - wv)[both] / np.maximum(np.abs(wv[both]), 1e-12)
res["step22_fits"] = int(both[0].sum())
res["step22_params_rel_diff"] = {"max": float(rel.max()), "p99.9": float(np.quantile(rel, 0.999)), "median": float(np.median(rel))}
res["step22_pure_nugget_equal"] = bool(np.array_equal(gv[2][both[2]] == 0, wv[2][both[2]] == 0))

# ---- step23: leave-one-out GWR series per (station, bandwidth, month), their MAE, and the bandwidth they choose -----------
t0 = time.perf_counter()
_, mae_a, bias_a, _ = xval.optim_nstns_anoms(stn, "tmin", stn_ids=ids)    # [12, nb, n]
res["step23_gpu_s"] = round(time.perf_counter() - t0, 2)
table23 = stn.stns.copy()
odb3 = orc.Db(sdb.StationSerialDataDb(table23, "tmin", stn.days, stn.var))
c3 = odb3.cols
msel = [np.nonzero(odb3.day_month == m)[0] for m in range(1, 13)]


def anom(q):
    j = idx[ids[q]]
    pt = orc.make_pt(c3["lon"][j], c3["lat"][j], c3["elev"][j], c3["tdi"][j], c3["lst"][:, j])
    o = odb3.obs[:, j].astype(np.float64)
    mae_o, bias_o = np.full((12, ladder.size), np.nan), np.full((12, ladder.size), np.nan)
    for x, k in enumerate(ladder):
        for m in range(1, 13):
            rc, series, _, _, _ = orc.gwr_mth(odb3, prm, pt, float(c3["norm"][m - 1, j]), m, int(k), excl=j, rm_zero_dist=True)
            if rc:
                return np.full((12, ladder.size), np.nan), np.full((12, ladder.size), np.nan)
            e = series - o[msel[m - 1]]
            mae_o[m - 1, x], bias_o[m - 1, x] = np.abs(e).mean(), e.mean()
    return mae_o, bias_o


t0 = time.perf_counter()
with ThreadPoolExecutor(nthr) as ex:
    parts = list(ex.map(anom, range(len(ids))))
want_mae = np.stack([p[0] for p in parts], axis=2)
want_bias = np.stack([p[1] for p in parts], axis=2)
res["step23_oracle_s"] = round(time.perf_counter() - t0, 1)
fa, fo = np.isfinite(mae_a), np.isfinite(want_mae)
res["step23_failed_equal"] = bool(np.array_equal(fa, fo))
both = fa & fo
res["step23_values"] = int(both.sum())
res["step23_mae_max_abs_diff_degC"] = float(np.abs(mae_a - want_mae)[both].max())
res["step23_bias_max_abs_diff_degC"] = float(np.abs(bias_a - want_bias)[both].max())
_, ch_g = xval.set_optim_nstns(table23.copy(), ids, mae_a, ladder, sdb.get_optim_anom_varname)
_, ch_o = xval.set_optim_nstns(table23.copy(), ids, want_mae, ladder, sdb.get_optim_anom_varname)
res["optim_nnghs_anom_choices"] = len(ch_g) * 12
res["optim_nnghs_anom_choices_differing"] = sum(int((ch_g[k] != ch_o[k]).sum()) for k in ch_g)

# ---- step24: leave-one-out normals + SE + daily values at every station with everything the farms set ------------------
xval.set_optim_nstns_tair_anom(stn, ids, mae_a)
t0 = time.perf_counter()
_, norms, se, dly, st = xval.xval_interp(stn, "tmin", stn_ids=ids, daily=True)
res["step24_gpu_s"] = round(time.perf_counter() - t0, 2)
odb4 = orc.Db(sdb.StationSerialDataDb(stn.stns.copy(), "tmin", stn.days, stn.var))
c4 = odb4.cols


def loo(q):
    j = idx[ids[q]]
    pt = orc.make_pt(c4["lon"][j], c4["lat"][j], c4["elev"][j], c4["tdi"][j], c4["lst"][:, j])
    rc, d, wn, ws = orc.interp(odb4, prm, pt, excl=j, rm_zero_dist=True, daily=True)
    if rc:
        return rc, 0.0, 0.0, 0.0
    return 0, float(np.abs(norms[q] - wn).max()), float(np.abs(se[q] - ws).max()), float(np.abs(dly[q] - d.astype(np.float32)).max())


t0 = time.perf_counter()
with ThreadPoolExecutor(nthr) as ex:
    r4 = list(ex.map(loo, range(len(ids))))
res["step24_oracle_s"] = round(time.perf_counter() - t0, 1)
rc4 = np.array([r[0] for r in r4])
res["step24_status_equal"] = bool(np.array_equal(rc4, st))
res["step24_failed"] = int((st != 0).sum())
okq = rc4 == 0
res["step24_normals_max_abs_diff_degC"] = float(max(r[1] for r, o in zip(r4, okq) if o))
res["step24_se_max_abs_diff_degC"] = float(max(r[2] for r, o in zip(r4, okq) if o))
res["step24_daily_f4_max_abs_diff_degC"] = float(max(r[3] for r, o in zip(r4, okq) if o))
res["step24_daily_values"] = int(okq.sum()) * int(stn.days.size)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "c5_parity.json"), "w"), indent=1)
print(json.dumps(res))
