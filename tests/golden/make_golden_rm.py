#!/usr/bin/env python3
"""Golden vectors for ``stns_rm`` given as an ARRAY of station ids (station_select.py:74-103: any number of ids, removed
with ``np.in1d``), made by EXECUTING the reference's own source slices -- the loader of make_golden.py (StationSelect,
KrigTair up to the R call, GwrTairAnom; the R / gstat call stands in as there).  Build container only:

    python tests/golden/make_golden_rm.py     ->  tests/golden/golden_rm_v1.npz

Cases: 2 ... 8 removed stations drawn from the point's 12 nearest (so that every one of them changes the neighbourhood), one list
holding an id that is not in the table (np.in1d ignores it), one list combined with rm_zero_dist_stns at a station's own
location; for each the selected indices / distances / weights of ``set_ngh_stns`` at k = 35 and 100, and
``KrigTair.krig`` / ``GwrTairAnom.gwr_mth`` of one month with the same list."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import sdb  # noqa: E402

MAXRM = 8


def main():
    geo, ss, it, opt = mg.load_reference()
    grid, tmin, tmax = mg.case_inputs()
    good = np.isnan(tmin.stns[sdb.BAD])
    ids_good = tmin.stns[sdb.STN_ID][good]
    id_to_idx = {s: i for i, s in enumerate(ids_good)}
    rng = np.random.default_rng(23)
    slct = ss["StationSelect"](tmin, good)
    slct_rmz = ss["StationSelect"](tmin, good, rm_zero_dist_stns=True)
    krig = it["KrigTair"](slct)
    gwr = it["GwrTairAnom"](slct)
    out = dict(input_hash=mg.input_hash(grid, tmin, tmax), lat=[], lon=[], k=[], rmz=[], excl=[], idx=[], dist=[], wgt=[],
               kr_cell=[], kr_mth=[], kr_excl=[], kr_mean=[], kr_var=[], gw_series=[])
    cells = rng.integers(5, 95, (7, 2))
    on_station = int(rng.integers(0, ids_good.size))
    for n_rm, (r, c) in zip(range(2, MAXRM + 1), cells):
        la, lo = grid["lat"][r], grid["lon"][c]
        s, rmz = slct, 0
        if n_rm == 5:                                            # this case sits ON a station, with rm_zero_dist_stns
            la, lo = tmin.stns[sdb.LAT][good][on_station], tmin.stns[sdb.LON][good][on_station]
            s, rmz = slct_rmz, 1
        s.set_ngh_stns(la, lo, 12, load_obs=False)
        near = list(s.ngh_stns[sdb.STN_ID])
        rm = [str(x) for x in rng.choice(near, n_rm, replace=False)]
        rm_arr = np.array(rm + (["NOT_A_STATION_ID"] if n_rm == 3 else []))       # np.in1d ignores a foreign id
        excl = np.full(MAXRM, -1, np.int32)
        excl[:n_rm] = sorted(id_to_idx[x] for x in rm)
        for k in (35, 100):
            s.set_ngh_stns(la, lo, k, load_obs=False, stns_rm=rm_arr)
            out["lat"].append(la); out["lon"].append(lo); out["k"].append(k); out["rmz"].append(rmz); out["excl"].append(excl)
            idx = np.full(100, -1, np.int32); idx[:k] = [id_to_idx[x] for x in s.ngh_stns[sdb.STN_ID]]
            d = np.zeros(100); d[:k] = s.ngh_dists
            w = np.zeros(100); w[:k] = s.ngh_wgt
            out["idx"].append(idx); out["dist"].append(d); out["wgt"].append(w)
        if rmz:
            continue
        # KrigTair.krig / GwrTairAnom.gwr_mth of one month with the same list (interp_tair.py:853-926, 261-314)
        pt = it["build_empty_pt"]()
        mg.fill_pt(pt, grid, r, c)
        mth = int(rng.integers(1, 13))
        for m in range(1, 13):
            pt[sdb.get_lst_varname(m)] = pt["tmin%02d" % m]
        mean, var = krig.krig(pt, mth, stns_rm=rm_arr)
        pt[sdb.get_norm_varname(mth)] = mean
        series = gwr.gwr_mth(pt, mth, stns_rm=rm_arr)
        out["kr_cell"].append((r, c)); out["kr_mth"].append(mth); out["kr_excl"].append(excl)
        out["kr_mean"].append(mean); out["kr_var"].append(var)
        out["gw_series"].append(np.asarray(series, np.float64))
    nd = max(a.size for a in out["gw_series"])
    out["gw_len"] = np.array([a.size for a in out["gw_series"]], np.int32)
    out["gw_series"] = np.array([np.pad(a, (0, nd - a.size)) for a in out["gw_series"]])
    np.savez_compressed(os.path.join(HERE, "golden_rm_v1.npz"), **{k: (np.array(v) if isinstance(v, list) else v) for k, v in out.items()})
    print("wrote golden_rm_v1.npz:", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
