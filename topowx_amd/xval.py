"""Config 5 (BASELINE.json configs[4]): the leave-one-out cross-validation farms and the bandwidth optimisation.

Counterpart of the reference's three MPI farms and of the reduction their writer rank runs at the end:

  step21_mpi_optim_nstns_norms.py:34-169   XvalTairNorm.run_xval per station -> |err| -> 'mae'[12, 16, nstn]
  step22_mpi_set_stn_variograms.py:33-142  StationKrigParams.get_krig_params per station -> vario_nug/psill/rngMM
  step23_mpi_optim_nstns_anoms.py:35-180   XvalTairAnom.run_xval per station -> mae / bias / r2
  step24_mpi_xval_interp.py:36-156         XvalTairOverall.run_interp per station -> daily + normals
  twx/interp/optimize.py:268-374           set_optim_nstns_tair_norm / _anom: mean MAE per climate division and
                                           month -> argmin over the bandwidth ladder -> optim_nnghsMM /
                                           optim_nnghs_anomMM of every station of the division

The reference deals ONE station at a time to a worker that loops over 16 bandwidths x 12 months in Python.  Here a
rank owns a strided share of the station list and pushes it through the batched C-ABI entries in chunks of a few
hundred stations: every (station, bandwidth, month) triple is one GPU point.  The statistics of step23 are reduced
on the device (``twx_gwr_xval_points``), so only three numbers per triple come back.  Ranks exchange results with
one ``all_gather`` of small arrays (``torch.distributed``: nccl = RCCL on the GPUs, gloo in the CPU tests); the data
path itself needs no collective -- stations are independent given the replicated table.

Layout choice: the reference's step23 hands its writer ``mae[n_bandwidths, 12]`` for a netCDF
variable shaped ``[12, n_bandwidths, nstn]``; this module always produces month-major ``[12, n_bandwidths, nstn]``,
which is what ``set_optim_nstns_tair_*`` index (``mae_climdiv[mth - 1, :, :]``, optimize.py:311).
"""
import os

import numpy as np

from .interp.optimize import XvalTairAnom, XvalTairNorm, XvalTairOverall, build_nstn_bandwidths
from .stationdb import (BAD, CLIMDIV, MASK, STN_ID, VARIO_NUG, VARIO_PSILL, VARIO_RNG, get_krigparam_varname,
                        get_optim_anom_varname, get_optim_varname)

__all__ = ["xval_station_ids", "shard", "optim_nstns_norms", "set_stn_variograms", "optim_nstns_anoms", "xval_interp", "set_optim_nstns",
           "set_optim_nstns_tair_norm", "set_optim_nstns_tair_anom", "write_optim_nstns_files",
           "set_optim_nstns_from_files", "run_config5", "DFLT_LADDER"]

DFLT_LADDER = build_nstn_bandwidths(35, 150, 0.10)        # step21:198, step23


def xval_station_ids(stn_da):
    """Stations that are cross-validated: inside the interpolation mask and not flagged bad
    (step21:146-149, step23 proc_coord, step24:135-137)."""
    s = stn_da.stns
    return s[STN_ID][np.isfinite(s[MASK]) & np.isnan(s[BAD])]


def shard(items, rank, world):
    """Strided share of ``items`` for ``rank`` (stations cost about the same: no balancing needed)."""
    return items[rank::world]


def _unshard(parts, n):
    """Inverse of ``shard`` along the last axis: parts[r] holds items r, r + world, ..."""
    world = len(parts)
    out = np.empty(parts[0].shape[:-1] + (n,), parts[0].dtype)
    for r, p in enumerate(parts):
        out[..., r::world] = p
    return out


def _all_gather(local, n_total, rank, world, device="cpu"):
    """Every rank's last-axis share -> the full array on every rank (one all_gather of padded blocks)."""
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    nmax = -(-n_total // world)
    pad = np.zeros(local.shape[:-1] + (nmax,), local.dtype)
    pad[..., :local.shape[-1]] = local
    t = torch.from_numpy(pad).to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    sizes = [len(range(r, n_total, world)) for r in range(world)]
    return _unshard([p.cpu().numpy()[..., :sizes[r]] for r, p in enumerate(parts)], n_total)


def _batches(ids, batch):
    for i in range(0, len(ids), batch):
        yield i, ids[i:i + batch]


KRIGALL_BYTES_PER_STATION = 22e9 / 1024        # device workspace of one cross-validated station x 16 bandwidths x 12 months (step21)


def pick_batch(ctx, bytes_per_station=KRIGALL_BYTES_PER_STATION, most=1024, least=32):
    """Stations per ``twx_krigall_points`` / ``twx_fit_vario_points`` call: as many as HALF the GPU's free memory holds
    (``twx_device_memory``), between ``least`` and ``most`` -- 1 024 on an MI355X by itself, fewer when ranks share a GPU or
    on a smaller one."""
    free, _ = ctx.device_memory()
    return int(max(least, min(most, (free // 2) // max(int(bytes_per_station), 1))))


def _run_batched(mine, batch, ctx, fn):
    """``fn(first_index, chunk)`` over ``mine`` in chunks of ``batch`` stations (None = ``pick_batch``); a chunk whose device
    workspace cannot be allocated is retried in halves instead of failing the run."""
    from ._lib import TwxError
    batch = (pick_batch(ctx) if ctx is not None else 256) if batch is None else int(batch)
    i = 0
    while i < len(mine):
        chunk = mine[i:i + batch]
        try:
            fn(i, chunk)
        except TwxError as e:
            if batch > 32 and ("hipMalloc" in str(e) or "ensure" in str(e) or "out of memory" in str(e).lower()):
                batch = max(32, batch // 2)
                continue
            raise
        i += len(chunk)
    return batch


def optim_nstns_norms(stn_da, tair_var, ladder=DFLT_LADDER, stn_ids=None, rank=0, world=1, batch=None, device=0,
                      gather_device="cpu"):
    """step21: leave-one-out error of the kriged normals for every station and bandwidth.

    Returns ``(stn_ids, mae)`` with ``mae[12, n_bandwidths, n_stations] = |interpolated - observed|``
    (step21:124-128); a station that cannot be cross-validated keeps NaN (the reference's worker prints the error
    and writes nothing: the netCDF fill value, masked in the reduction).  ``batch`` stations (x 16 bandwidths x 12 months
    points, ~22 GB of workspace at 1 024) go through ``twx_krigall_points`` per call: 256 -> 1 024 took step21 from 0.53 to
    0.48 s on the 12 000-station database (tests/tools/gpu_c5_batch.py; the per-call host work amortises).  Default (None):
    as many as half the GPU's free memory holds, at most 1 024 (``pick_batch``); a chunk that still cannot be allocated is
    retried in halves."""
    ids = xval_station_ids(stn_da) if stn_ids is None else np.asarray(stn_ids)
    mine = shard(ids, rank, world)
    ladder = np.asarray(ladder)
    mae = np.full((12, ladder.size, len(mine)), np.nan)
    xv = XvalTairNorm(stn_da, tair_var, device=device)
    try:
        def one(i, chunk):
            err, ok = xv.run_xval_many(chunk, ladder, raise_on_error=False)          # [ns, 12, nb]
            blk = np.abs(np.transpose(err, (1, 2, 0)))
            blk[:, :, ~ok] = np.nan
            mae[:, :, i:i + len(chunk)] = blk
        _run_batched(mine, batch, getattr(xv, "ctx", None), one)
    finally:
        xv.close()
    return ids, _all_gather(mae, len(ids), rank, world, gather_device)


def set_stn_variograms(stn_da, tair_var, stn_ids=None, rank=0, world=1, batch=None, device=0, gather_device="cpu"):
    """step22: the variogram parameters of every station location for every month, fitted with the station's optimised
    bandwidth (``optim_nnghsMM`` smoothed over its neighbours; the station stays inside its own neighbourhood,
    interp_tair.py:667,681) and written into ``vario_nugMM / vario_psillMM / vario_rngMM`` of ``stn_da.stns`` for the
    stations inside the mask that are not flagged bad (step22:33-66 worker, :68-118 writer, :119-121 station list).

    Returns ``(stn_ids, nug[12, n], psill[12, n], rng[12, n])``; a station whose fit fails keeps NaN in all twelve
    months (the reference writes the f8 fill value, which ``_build_stn_struct`` reads back as NaN,
    station_data.py:159-164).  Every rank fits its strided share and all ranks end with the same table."""
    from .interp.optimize import StationKrigParams
    ids = xval_station_ids(stn_da) if stn_ids is None else np.asarray(stn_ids)
    mine = shard(ids, rank, world)
    out = np.full((3, 12, len(mine)), np.nan)
    kp = StationKrigParams(stn_da, tair_var, device=device)
    try:
        def one(i, chunk):
            nug, psill, rng, ok = kp.get_krig_params_many(chunk, raise_on_error=False)     # [ns, 12]
            for q, a in enumerate((nug, psill, rng)):
                blk = a.T.copy()
                blk[:, ~ok] = np.nan
                out[q, :, i:i + len(chunk)] = blk
        _run_batched(mine, batch, getattr(kp, "ctx", None), one)
    finally:
        kp.close()
    out = _all_gather(out, len(ids), rank, world, gather_device)
    stns = stn_da.stns
    pos = {s: i for i, s in enumerate(stns[STN_ID])}
    rows = np.array([pos[s] for s in ids], np.int64)
    for m in range(1, 13):                                                                   # step22:104-110
        stns[get_krigparam_varname(m, VARIO_NUG)][rows] = out[0, m - 1]
        stns[get_krigparam_varname(m, VARIO_PSILL)][rows] = out[1, m - 1]
        stns[get_krigparam_varname(m, VARIO_RNG)][rows] = out[2, m - 1]
    return ids, out[0], out[1], out[2]


def optim_nstns_anoms(stn_da, tair_var, ladder=DFLT_LADDER, stn_ids=None, rank=0, world=1, batch=128, device=0,
                      gather_device="cpu"):
    """step23: leave-one-out GWR anomalies for every station, bandwidth and month.

    Returns ``(stn_ids, mae, bias, r2)``, each ``[12, n_bandwidths, n_stations]`` (month-major, see the module
    docstring); NaN where the reference's worker would have stored the fill value (step23:60-66)."""
    ids = xval_station_ids(stn_da) if stn_ids is None else np.asarray(stn_ids)
    mine = shard(ids, rank, world)
    ladder = np.asarray(ladder)
    out = [np.full((12, ladder.size, len(mine)), np.nan) for _ in range(3)]
    xv = XvalTairAnom(stn_da, tair_var, device=device)
    try:
        for i, chunk in _batches(mine, batch):
            bias, mae, r2, ok = xv.run_xval_many(chunk, ladder, raise_on_error=False)   # [ns, nb, 12]
            for dst, src in zip(out, (mae, bias, r2)):
                blk = np.transpose(src, (2, 1, 0)).copy()
                blk[:, :, ~ok] = np.nan
                dst[:, :, i:i + len(chunk)] = blk
    finally:
        xv.close()
    mae, bias, r2 = (_all_gather(a, len(ids), rank, world, gather_device) for a in out)
    return ids, mae, bias, r2


def xval_interp(stn_da, tair_var, stn_ids=None, daily=True, rank=0, world=1, batch=512, device=0,
                gather_device="cpu"):
    """step24: leave-one-out interpolation of normals (+ daily values) at every station.

    Returns ``(stn_ids, norms[n, 12], se[n, 12], daily[n, ndays] float32 or None, status[n])``; failed stations
    keep NaN (the reference writes fill values, step24:52-62).  Daily values are gathered as float32, the dtype
    the reference's output database stores (step24:30-33)."""
    ids = xval_station_ids(stn_da) if stn_ids is None else np.asarray(stn_ids)
    mine = shard(ids, rank, world)
    nd = stn_da.days.size
    norms = np.full((12, len(mine)), np.nan)
    se = np.full((12, len(mine)), np.nan)
    st = np.zeros((1, len(mine)), np.int32)
    dly = np.full((nd, len(mine)), np.nan, np.float32) if daily else None
    xv = XvalTairOverall(stn_da, tair_var, device=device)
    try:
        for i, chunk in _batches(mine, batch):
            d, n, s, code = xv.run_interp_many(chunk, daily=daily, raise_on_error=False)
            norms[:, i:i + len(chunk)] = n.T
            se[:, i:i + len(chunk)] = s.T
            st[0, i:i + len(chunk)] = code
            if daily:
                dly[:, i:i + len(chunk)] = d.T.astype(np.float32)
    finally:
        xv.close()
    norms = _all_gather(norms, len(ids), rank, world, gather_device).T
    se = _all_gather(se, len(ids), rank, world, gather_device).T
    st = _all_gather(st, len(ids), rank, world, gather_device)[0]
    if daily:
        dly = _all_gather(dly, len(ids), rank, world, gather_device).T
    return ids, norms, se, dly, st


def set_optim_nstns(stns, stn_ids, mae, ladder, namer):
    """optimize.py:268-374 on arrays: for every climate division and month the bandwidth with the smallest MAE
    averaged over the division's cross-validated stations becomes ``namer(mth)`` of EVERY station of the division
    (``climdiv_mask`` is taken over the whole table, :308).  Failed stations (NaN) are left out of the mean, as the
    masked fill values are in the reference.  ``stns`` is modified in place and returned together with the chosen
    bandwidths ``{climdiv: [12]}``."""
    ladder = np.asarray(ladder)
    ids = np.asarray(stn_ids)
    pos = {s: i for i, s in enumerate(stns[STN_ID])}
    div_of_xval = stns[CLIMDIV][[pos[s] for s in ids]]
    climdiv_stns = stns[CLIMDIV]
    # The reference loops over ~340 division files; the synthetic CONUS table has ~2 000 divisions, and 24 masked-array
    # reductions per division took longer than the GPU farm that produced the errors.  Here: the cross-validated stations
    # are ordered by division once, and a division's twelve months are reduced together on plain arrays -- the same row
    # sums (numpy's pairwise summation along the contiguous station axis), counts and first-minimum choice as the
    # masked mean of :312-313, so also the same choice when two bandwidths tie.
    order = np.argsort(div_of_xval, kind="stable")
    order = order[np.isfinite(div_of_xval[order])]
    xdivs, start = np.unique(div_of_xval[order], return_index=True)              # divisions with an MAE "file" (step21:96-104)
    stop = np.append(start[1:], order.size)
    chosen = {}
    if xdivs.size:
        m = np.asarray(mae, np.float64)[:, :, order]                             # [12, nb, xval stations by division]
        fin = np.isfinite(m)
        m0 = np.where(fin, m, 0.0)
        picks = np.empty((12, xdivs.size), ladder.dtype)
        for q in range(xdivs.size):
            seg = slice(start[q], stop[q])
            cnt = fin[:, :, seg].sum(axis=2)
            with np.errstate(invalid="ignore", divide="ignore"):
                mmae = m0[:, :, seg].sum(axis=2) * 1. / cnt                      # :312 (no finite error: masked there)
            mmae[cnt == 0] = np.inf                                              # (a masked mean never wins the argmin)
            picks[:, q] = ladder[np.argmin(mmae, axis=1)]                        # :313 (first minimum)
        # :308/:314 every station of the division, cross-validated or not
        fin_all = np.nonzero(np.isfinite(climdiv_stns))[0]
        slot = np.minimum(np.searchsorted(xdivs, climdiv_stns[fin_all]), xdivs.size - 1)
        hit = xdivs[slot] == climdiv_stns[fin_all]
        rows, slot = fin_all[hit], slot[hit]
        for mth in range(1, 13):
            stns[namer(mth)][rows] = picks[mth - 1][slot]
        for q, clim_div in enumerate(xdivs):
            chosen[float(clim_div)] = picks[:, q].copy()
    return stns, chosen


def write_optim_nstns_files(path_out, stn_da, stn_ids, mae, ladder=DFLT_LADDER):
    """The side product of the step21 / step23 writer rank (step21:83-128, optimize.py:39-82): one
    ``optim_nstns_<var>_climdiv<id>.nc`` per climate division holding the MAE cube ``[12, n_bandwidths, stations of
    the division]`` of the cross-validated stations (through ``topowx_amd.ncio``: NetCDF-4, or classic netCDF without libhdf5).  Returns the paths."""
    import os
    from . import ncio
    os.makedirs(path_out, exist_ok=True)
    stns = stn_da.stns
    ids = np.asarray(stn_ids)
    pos = {s: i for i, s in enumerate(stns[STN_ID])}
    div_of_xval = stns[CLIMDIV][[pos[s] for s in ids]]
    paths = []
    for clim_div in np.unique(div_of_xval[np.isfinite(div_of_xval)]):       # step21:83 loops the divisions of the xval stations
        cols = np.nonzero(div_of_xval == clim_div)[0]
        paths.append(ncio.write_climdiv_optim_nstns_db(path_out, stn_da.var_name, ids[cols], ladder, clim_div,
                                                       np.asarray(mae)[:, :, cols]))
    return paths


def set_optim_nstns_from_files(stn_da, path_xval_ds, namer, strict=True):
    """``set_optim_nstns_tair_norm / _anom`` as the reference runs them (optimize.py:285-316, :339-370): from the
    per-division MAE files under ``path_xval_ds``.  A division without a file raises ``IOError`` -- the reference
    fails at ``Dataset(fpath)`` (optimize.py:302-304), and stations left with stale / NaN ``optim_nnghs`` would only
    surface later as TWX_CELL_NNGHS failures far from the cause; ``strict=False`` skips such divisions and returns
    them under the key ``"missing"``.  Same arithmetic as ``set_optim_nstns`` (masked mean over the division's
    stations, first minimum)."""
    import os
    from . import ncio
    stns = stn_da.stns
    climdiv_stns = stns[CLIMDIV]
    chosen = {}
    for clim_div in np.unique(climdiv_stns[np.isfinite(climdiv_stns)]):       # :300
        fpath = ncio.climdiv_optim_nstns_path(path_xval_ds, stn_da.var_name, clim_div)
        if not os.path.exists(fpath):
            if strict:
                raise IOError("no cross-validation file for climate division %g: %s" % (clim_div, fpath))
            chosen.setdefault("missing", []).append(float(clim_div))
            continue
        mae_climdiv, nnghs_climdiv, _ = ncio.read_climdiv_optim_nstns_db(fpath)  # :304-307
        climdiv_mask = np.nonzero(climdiv_stns == clim_div)[0]                   # :308
        pick = np.zeros(12, nnghs_climdiv.dtype)
        for mth in range(1, 13):
            mmae = np.ma.mean(np.ma.masked_invalid(mae_climdiv[mth - 1]), axis=1)    # :311-312
            min_idx = int(np.argmin(mmae))                                       # :313
            stns[namer(mth)][climdiv_mask] = nnghs_climdiv[min_idx]              # :314
            pick[mth - 1] = nnghs_climdiv[min_idx]
        chosen[float(clim_div)] = pick
    return chosen


def _set_optim_files(stn_da, path_xval_ds, namer, long_name):
    """The reference's route (optimize.py:285-316 / :339-370): ``add_stn_variable`` for the twelve months, then the
    per-division files; with a database opened ``mode='r+'`` the columns reach its file."""
    cols = {}
    if hasattr(stn_da, "add_stn_variable"):
        for mth in range(1, 13):
            cols[mth] = stn_da.add_stn_variable(namer(mth), long_name % mth, "", "f8")   # (reset, as :292-295)
    chosen = set_optim_nstns_from_files(stn_da, path_xval_ds, namer)
    for mth, v in cols.items():
        v[:] = stn_da.stns[namer(mth)]                         # write-through of what the loop set
    if getattr(stn_da, "ds", None) is not None:
        stn_da.ds.sync()                                       # optimize.py:316
    return chosen


def set_optim_nstns_tair_norm(stn_da, stn_ids, mae=None, ladder=DFLT_LADDER):
    """optimize.py:268-316 (the end of step21): writes ``optim_nnghsMM``.  Two call shapes: the reference's
    ``set_optim_nstns_tair_norm(stnda, path_xval_ds)`` -- from the per-division MAE files under a directory -- and the
    array form ``(stn_da, stn_ids, mae[12, nb, n], ladder)`` the GPU farms use."""
    if mae is None and isinstance(stn_ids, (str, os.PathLike)):
        return _set_optim_files(stn_da, os.fspath(stn_ids), get_optim_varname,
                                "Optimal number of neighbors to use for monthly normal interpolation for month %d")
    return set_optim_nstns(stn_da.stns, stn_ids, mae, ladder, get_optim_varname)[1]


def set_optim_nstns_tair_anom(stn_da, stn_ids, mae=None, ladder=DFLT_LADDER):
    """optimize.py:318-374 (the end of step23): writes ``optim_nnghs_anomMM``; same two call shapes."""
    if mae is None and isinstance(stn_ids, (str, os.PathLike)):
        return _set_optim_files(stn_da, os.fspath(stn_ids), get_optim_anom_varname,
                                "Optimal number of neighbors to use for daily anomaly interpolation for month %d")
    return set_optim_nstns(stn_da.stns, stn_ids, mae, ladder, get_optim_anom_varname)[1]


def config5_bbox(db):
    """Bounding box the synthetic station database of ``run_config5`` is drawn over: "c5" = the CONUS-shaped grid of
    BASELINE.json configs[2..4] (12 000 stations per variable, seed 2), "c2" = the C2 tile."""
    from . import synth
    nrows, ncols, lat_north, lon_west, _, seed = synth.CONFIGS["C3" if db == "c5" else "C2"]
    lat = lat_north - (np.arange(nrows) + 0.5) * synth.CELL
    lon = lon_west + (np.arange(ncols) + 0.5) * synth.CELL
    return (lat.min(), lat.max(), lon.min(), lon.max()), seed


def run_config5(nstns=2000, years=3, var="tmin", max_stations=0, rank=0, world=1, device=0, gather_device="cpu", db="c2",
                stn_path=None):
    """BASELINE.json configs[4] end to end on a synthetic database, timed, in the order the reference runs the farms:
    step21 (variogram fit + kriging per (station, bandwidth, month), step21:34-64), ``set_optim_nstns_tair_norm``,
    step22 (every station's variogram with the optimised bandwidths, step22:33-66), step23 (GWR series + statistics,
    step23:35-70), ``set_optim_nstns_tair_anom``, step24 (normals + daily values with everything the previous steps
    set, step24:36-69).  ``db="c5"``: SURVEY 8d's configs[4] database (stations over the CONUS-shaped grid, seed 2;
    ``nstns`` = 12 000 there); ``db="c2"``: the C2 tile's.  ``stn_path``: with ``world`` > 1 rank 0 builds the
    database once and saves it there, the other ranks load it after the barrier (N x setup otherwise).
    Returns (timings / rates dict, arrays dict for spot checks)."""
    import time
    import datetime as dt
    from . import synth
    from .dates import get_days_metadata
    from .stationdb import StationDataWrkChk
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1980 + years, 12, 31))
    bbox, seed = config5_bbox(db)
    t0 = time.perf_counter()
    if stn_path and world > 1:
        import torch.distributed as dist
        if rank == 0:
            synth.make_stations(bbox, nstns, seed, var, days, with_obs=True).save(stn_path)
        dist.barrier()
        stn = StationDataWrkChk.load(stn_path)
    else:
        stn = synth.make_stations(bbox, nstns, seed, var, days, with_obs=True)
    ids = xval_station_ids(stn)
    if max_stations:
        ids = ids[:max_stations]
    res = {"stations": int(len(ids)), "stations_in_db": int(stn.stns.size), "db": db, "ladder": int(DFLT_LADDER.size),
           "days": int(days.size), "n_gpus": world, "setup_s": time.perf_counter() - t0}
    kw = dict(stn_ids=ids, rank=rank, world=world, device=device, gather_device=gather_device)
    t0 = time.perf_counter()
    _, mae_n = optim_nstns_norms(stn, var, **kw)
    res["step21_s"] = time.perf_counter() - t0
    stn_before = stn.stns.copy()                     # the table step21 cross-validated against (for spot checks)
    t0 = time.perf_counter()
    set_optim_nstns_tair_norm(stn, ids, mae_n)
    res["set_optim_s"] = time.perf_counter() - t0    # (host: the reductions between the farms, norm + anom)
    t0 = time.perf_counter()
    _, nug, psill, rng = set_stn_variograms(stn, var, **kw)
    res["step22_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, mae_a, _, _ = optim_nstns_anoms(stn, var, **kw)
    res["step23_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    set_optim_nstns_tair_anom(stn, ids, mae_a)
    res["set_optim_s"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    _, norms, _, _, st = xval_interp(stn, var, daily=True, **kw)
    res["step24_s"] = time.perf_counter() - t0
    res["step21_systems_per_s"] = len(ids) * DFLT_LADDER.size * 12 / res["step21_s"]
    res["step22_fits_per_s"] = len(ids) * 12 / res["step22_s"]
    res["step23_series_per_s"] = len(ids) * DFLT_LADDER.size * 12 / res["step23_s"]
    res["step24_station_days_per_s"] = len(ids) * days.size / res["step24_s"]
    res["failed"] = int((st != 0).sum())
    res["step21_mae_finite_frac"] = float(np.isfinite(mae_n).mean())
    res["step22_fitted_frac"] = float(np.isfinite(nug).mean())
    res["step22_pure_nugget_frac"] = float((rng[np.isfinite(rng)] == 0).mean()) if np.isfinite(rng).any() else 0.0
    return res, {"ids": ids, "mae_norm": mae_n, "mae_anom": mae_a, "norms": norms, "status": st, "stn": stn,
                 "stns_step21": stn_before, "vario": (nug, psill, rng)}


def main():
    """All three farms on a synthetic database, timed: ``python -m topowx_amd.xval --nstns 10000``
    (under torchrun every rank takes its share)."""
    import argparse
    import json
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--nstns", type=int, default=2000)
    ap.add_argument("--years", type=int, default=3)
    ap.add_argument("--var", default="tmin")
    ap.add_argument("--max-stations", type=int, default=0, help="cross-validate only the first N stations (0 = all)")
    ap.add_argument("--db", choices=("c2", "c5"), default="c2", help="c5: stations over the CONUS-shaped grid, seed 2 (use --nstns 12000)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gdev = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        gdev = "cuda:%d" % local
    res, _ = run_config5(args.nstns, args.years, args.var, args.max_stations, rank, world, local, gdev, db=args.db)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
