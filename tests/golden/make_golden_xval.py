#!/usr/bin/env python3
"""Golden vectors of the cross-validation callers (BASELINE.json configs[4]), made by
EXECUTING the reference's source slices (build container only; needs /root/reference):

    python tests/golden/make_golden_xval.py

Slices executed on top of the ones ``make_golden.load_reference`` loads (StationSelect, GwrTairAnom,
_gwr_series ...):
  twx/interp/optimize.py:476-545   XvalTairAnom (``run_xval``: 16 x 12 ``gwr_mth(stn, mth, nnghs,
                                   stns_rm=id)`` + bias / MAE / r^2); ``__init__`` opens a netCDF file and
                                   is bypassed: the object gets the same ``stn_da`` / ``gwr`` members from
                                   the in-memory synthetic database
  twx/interp/optimize.py:268-374   set_optim_nstns_tair_norm / set_optim_nstns_tair_anom (mean MAE per
                                   climate division and month -> argmin -> optim_nnghsMM of its stations),
                                   run against in-memory stand-ins for ``netCDF4.Dataset`` (returns the MAE
                                   cube of one division), ``StatusCheck`` and ``stnda.add_stn_variable``
                                   (returns a plain array that is assigned into)

Environment note: the reference ran on a pre-NEP-50 numpy where ``float32_array - float64_scalar`` stays
float32 (optimize.py:525); numpy 2 promotes to float64.  The observed anomalies therefore differ by one f4
rounding (<= 2e-6 degC) from what the original stack would have produced -- far inside the 1e-4 degC bar.
No reference text is stored in the fixture: inputs hash + expected outputs only.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from topowx_amd import stationdb as sdb  # noqa: E402

LADDER_STATIONS = (5, 123, 287)          # indices into the good stations
GWR_PROBES = ((5, 35, 1), (5, 147, 7), (123, 57, 3), (287, 101, 12), (287, 39, 6))   # (station, nnghs, month)


def synthetic_mae(n_stn, seed):
    """MAE cube [12, 16, n_stn] of one climate division with a known, non-trivial argmin per month."""
    rng = np.random.default_rng(seed)
    base = np.abs(rng.normal(1.0, 0.25, (12, 16, n_stn)))
    dip = rng.integers(0, 16, 12)
    for m in range(12):
        base[m, dip[m]] *= 0.55
    # one month with an exact tie between two bandwidths: np.argmin keeps the first
    base[4, 9] = base[4, 3]
    return base


def main():
    geo, ss, it, opt = mg.load_reference()
    np.seterr(all="raise", under="ignore")
    grid, tmin, tmax = mg.case_inputs()
    out = dict(input_hash=mg.input_hash(grid, tmin, tmax))
    import scipy.stats as stats

    # ---- XvalTairAnom.run_xval (optimize.py:505-545) ------------------------------------------
    ns = dict(np=np, stats=stats, StationSelect=ss["StationSelect"], GwrTairAnom=it["GwrTairAnom"],
              StationSerialDataDb=None, BAD=sdb.BAD, STN_ID=sdb.STN_ID, get_norm_varname=sdb.get_norm_varname)
    exec(compile(mg._slice("twx/interp/optimize.py", 476, 545), "optimize_476", "exec"), ns)
    good = np.isnan(tmin.stns[sdb.BAD])
    xa = object.__new__(ns["XvalTairAnom"])
    xa.stn_da = tmin
    xa.gwr = it["GwrTairAnom"](ss["StationSelect"](tmin, stn_mask=good, rm_zero_dist_stns=True))   # :499-500
    ladder = opt["build_nstn_bandwidths"](35, 150, 0.10)
    ids = tmin.stns[sdb.STN_ID][good]
    bias, mae, r2 = [], [], []
    for j in LADDER_STATIONS:
        b, m, r = xa.run_xval(ids[j], ladder)
        bias.append(b); mae.append(m); r2.append(r)
    out.update(xa_stn=np.array(LADDER_STATIONS, np.int32), xa_ladder=ladder.astype(np.int32),
               xa_bias=np.array(bias), xa_mae=np.array(mae), xa_r2=np.array(r2))
    # raw gwr_mth(stn, mth, nnghs, stns_rm=id) series (the twx_gwr_points call shape of step23)
    probes, series = [], []
    nd_max = max(v.size for v in tmin.mth_idx.values())
    for j, k, m in GWR_PROBES:
        rec = tmin.stns[good][j]
        s = xa.gwr.gwr_mth(rec, m, k, stns_rm=rec[sdb.STN_ID])
        row = np.full(nd_max, np.nan)
        row[:s.size] = s
        probes.append((j, k, m)); series.append(row)
    out.update(gx_probe=np.array(probes, np.int32), gx_series=np.array(series))

    # ---- set_optim_nstns_tair_norm / _anom (optimize.py:268-374) ---------------------------------
    stns = tmin.stns.copy()
    # three synthetic climate divisions + stations without one (NaN -> skipped by np.isfinite, :300)
    rng = np.random.default_rng(77)
    div = rng.choice([101.0, 102.0, 4407.0, np.nan], stns.size, p=[0.4, 0.3, 0.2, 0.1])
    stns[sdb.CLIMDIV] = div
    cubes = {}
    for d in (101, 102, 4407):
        cubes[d] = synthetic_mae(int((div == d).sum()), 900 + d)

    class FakeVar(object):
        def __init__(self, a):
            self.a = a

        def __getitem__(self, key):
            return self.a[key]

    class FakeDataset(object):
        def __init__(self, fpath, *a, **k):
            d = int(os.path.basename(fpath).split("climdiv")[1].split(".")[0])
            assert os.path.basename(fpath) == "optim_nstns_%s_climdiv%d.nc" % ("tmin", d)
            self.variables = {"mae": FakeVar(cubes[d]), "min_nghs": FakeVar(ladder.astype(np.float64))}

    class FakeStnDa(object):
        def __init__(self):
            self.stns = stns
            self.var_name = "tmin"
            self.added = {}
            self.ds = types.SimpleNamespace(sync=lambda: None)

        def add_stn_variable(self, name, long_name, units, dtype, fill_value=None):
            self.added[name] = np.full(stns.size, fill_value, np.float64)
            return self.added[name]

    class FakeStatus(object):
        def __init__(self, *a):
            pass

        def increment(self, *a):
            pass

    fake_nc4 = types.SimpleNamespace(default_fillvals={"f8": 9.969209968386869e36})
    rn = dict(np=np, os=os, Dataset=FakeDataset, netCDF4=fake_nc4, StatusCheck=FakeStatus, CLIMDIV=sdb.CLIMDIV,
              get_optim_varname=sdb.get_optim_varname, get_optim_anom_varname=sdb.get_optim_anom_varname)
    exec(compile(mg._slice("twx/interp/optimize.py", 268, 374), "optimize_268", "exec"), rn)
    da = FakeStnDa()
    rn["set_optim_nstns_tair_norm"](da, "/nowhere")
    optim = np.stack([da.added[sdb.get_optim_varname(m)] for m in range(1, 13)])
    da2 = FakeStnDa()
    rn["set_optim_nstns_tair_anom"](da2, "/nowhere")
    optim_anom = np.stack([da2.added[sdb.get_optim_anom_varname(m)] for m in range(1, 13)])
    out.update(so_climdiv=div, so_divs=np.array([101, 102, 4407], np.int32),
               so_mae_101=cubes[101], so_mae_102=cubes[102], so_mae_4407=cubes[4407],
               so_optim=optim, so_optim_anom=optim_anom, so_fill=np.float64(9.969209968386869e36))

    path = os.path.join(HERE, "golden_xval_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
