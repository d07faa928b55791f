#!/usr/bin/env python3
"""Benchmark of the interpolation hot path on MI355X.

Headline: one "step" = one pass of the hot path (twx_interp_grid_dev) over one synthetic 250x250 30-arcsec
tile with ~10k stations, 12 monthly Tmin normals + standard errors (BASELINE.json configs[1]); inputs are
resident in HBM before the timed region.  With N > 1 every rank interpolates its OWN tile against a replicated
station table (tiles partition embarrassingly): weak scaling, no data-path collective.

``python bench.py --gpus N`` without a torchrun environment starts N ranks itself (child processes through
``python -m torch.distributed.run``, spawned before this process touches the GPU) and relays rank 0's line;
under torchrun (WORLD_SIZE set) it is a rank.

Rank 0 prints ONE JSON line (contract in the task description) carrying
  roofline      dominant kernels = the universal-kriging launches, HIP-event timed inside the library on the
                launch stream; HBM as BASELINE.json asks, the binding fp64-vector figure next to it (``fp64``)
  daily         (N = 1) the cell-DAY producing path, timed the same way: the same tile, Tmin + Tmax, normals +
                GWR + 10 years of daily int16 values + Tmin>=Tmax fixer, outputs resident in HBM
  cpu_baseline  (N = 1) the CPU oracle on bounded samples of the headline workload: all host cores (>= 64 cells
                per thread) and one core
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL_MONTH = 13.1   # Tmin-only normals, 61 B in + 96 B out per cell / 12
ALG_BYTES_PER_CELL_DAY = 2.03     # int16 out + amortised inputs / observation matrix
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VEC_PEAK_TFLOPS = 78.6       # vendor fp64 vector peak


def uk_flops(k):
    """Algorithmic fp64 flops of one (cell, month) kriging system."""
    k = np.asarray(k, np.float64)
    p = 4
    return k ** 3 / 3.0 + (p + 3) * k ** 2 + 60.0 * k * (k - 1) / 2.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=250, help="tile edge in cells (default: the C2 tile)")
    ap.add_argument("--nstns", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-daily", action="store_true", help="skip the daily (cell-days) record")
    ap.add_argument("--daily-years", type=int, default=10, help="years of days of the daily record (1981-...)")
    ap.add_argument("--stream-tiles", type=int, default=4, help="tiles of the streamed (PCIe-inclusive) daily record; 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=0, help="edge of the all-core CPU sample window (0 = auto)")
    return ap.parse_args()


def spawn(args):
    """--gpus N outside torchrun: start N ranks as children.  Nothing in this process has touched the GPU
    (device_count() does not initialise it), and it never re-executes itself: it waits and exits with the
    children's code."""
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        # fewer GPUs than ranks (1-GPU box): control-flow run, all ranks share GPU 0, rendezvous over gloo
        env["TWX_BENCH_SHARE_GPU"] = "1"
        env["TWX_BENCH_BACKEND"] = "gloo"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def latest_traffic():
    """HBM bytes per kriging launch from the newest committed PMC reduction (profiles/r*_bench_hbm_traffic.json)."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_hbm_traffic.json")))
    if not paths:
        return None, None
    tj = json.load(open(paths[-1]))
    traffic = tj["FETCH_SIZE"]["k_uk_per_launch_bytes"] + tj["WRITE_SIZE"]["k_uk_per_launch_bytes"]
    src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this workload, KB units x "
           "1024, per kriging launch; 4-byte loads of the fp32 pair-distance cache: the guide's x2 correction for "
           "16-B/lane streaming reads does not apply)" % os.path.basename(paths[-1]))
    return traffic, src


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(args)

    import torch
    import torch.distributed as dist
    from topowx_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TWX_BENCH_BACKEND=gloo + TWX_BENCH_SHARE_GPU=1: control-flow check of the N > 1 path on a 1-GPU box
    backend = os.environ.get("TWX_BENCH_BACKEND", "nccl")
    shared = os.environ.get("TWX_BENCH_SHARE_GPU") == "1"
    if shared:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- synthetic workload -----------------------------------------------------------------
    # Every rank holds the SAME replicated station table (the N = 1 table) and interpolates its own
    # 250x250 tile: rank r's tile is the C2 tile shifted by multiples of 1/8 degree, so all tiles lie
    # inside the station region and carry statistically the same work (weak scaling, fixed per-GPU work).
    Y = X = args.size
    base = synth.make_grid("C2", nrows=Y, ncols=X)
    stn = synth.make_stations(base["bbox"], args.nstns, synth.CONFIGS["C2"][5], "tmin")
    if rank == 0:
        grid = base
    else:
        grid = synth.make_grid("C2", nrows=Y, ncols=X, lon_west=-111.0 + 0.125 * (rank % 4),
                               lat_north=46.0 - 0.125 * (rank // 4))

    ctx = _lib.Context(device=local)
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    a = ctx.grid_arrays(grid)
    d_in = {k: up(v) for k, v in a.items()}
    d_norm = torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
    d_se = torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
    d_ninv = torch.full((Y, X), int(_lib.FILL_I4), dtype=torch.int32, device=dev)
    d_stat = torch.full((Y, X), -1, dtype=torch.int32, device=dev)
    g = _lib.TwxGrid(Y, X, d_in["mask"].data_ptr(), d_in["lat"].data_ptr(), d_in["lon"].data_ptr(),
                     d_in["elev"].data_ptr(), d_in["tdi"].data_ptr(), d_in["climdiv"].data_ptr(),
                     d_in["lst_night"].data_ptr(), d_in["lst_day"].data_ptr())
    o = _lib.TwxGridOut(d_norm.data_ptr(), d_se.data_ptr(), None, None, None, None, d_ninv.data_ptr(),
                        d_stat.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        """W untimed + K timed steps between barrier + synchronize; max over ranks; per-step kernel timings."""
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        kern = []
        for _ in range(args.steps):
            step()
            kern.append(ctx.timing())       # HIP events on the launch stream (synchronises this step)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, kern

    elapsed, kern = timed(lambda: ctx.interp_grid_dev(g, o, _lib.VAR_TMIN_BIT, stream))
    status = d_stat.cpu().numpy()
    ncell_ok = int((status == 0).sum())
    units_per_step = ncell_ok * 12                     # (cell, month) outputs, each mean + SE
    value = world * units_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernels (universal kriging) ---------------------------------
    uk_ms = float(np.mean([t["uk_ms"] for t in kern]))
    launches = max(1, int(kern[-1]["uk_launches"]))
    solves = int(kern[-1]["uk_solves"])
    ach_gbs = ALG_BYTES_PER_CELL_MONTH * solves / (uk_ms * 1e-3) / 1e9
    # bandwidths actually used by the timed steps (diagnostic accessor, no extra launches)
    ks = ctx.last_bandwidths(_lib.TMIN).ravel()
    flops_per_solve = float(uk_flops(ks[ks > 0]).mean())
    # systems per kriging launch: matrix rows (k + 8, rounded up to the kernel's size) -> count
    edges = np.array([40, 48, 56, 64, 72, 80, 88, 96, 112, 128, 144, 160])
    rows_hist = np.bincount(np.searchsorted(edges, ks[ks > 0] + 8), minlength=edges.size + 1)
    ach_tflops = flops_per_solve * solves / (uk_ms * 1e-3) / 1e12
    # PMC counters cannot be read from inside the process: the committed measurement of THIS workload is quoted
    traffic, traffic_src = (None, None)
    if world == 1 and args.size == 250 and args.nstns == 10000:
        traffic, traffic_src = latest_traffic()

    res = {
        "metric": "grid-cell-days interpolated/sec",
        "value": value,
        "unit": "cell-months/s (normals config: one time step = one calendar month, mean + SE)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2: one %dx%d 30-arcsec tile per GPU, %d synthetic stations, 12 monthly Tmin "
                               "normals + SE (BASELINE.json configs[1])" % (Y, X, ctx.nstn[_lib.TMIN]),
                   "cells_ok": ncell_ok, "mean_nnghs": float(ks[ks > 0].mean()),
                   "parallelism": "tiles partitioned over %d GPU(s), station table replicated" % world},
        "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": ALG_BYTES_PER_CELL_MONTH * solves / launches,
                     "kernel": "universal kriging: k_tile_dist + k_ukw<..> + k_uk<..> (%d launches per step)" % launches,
                     "kernel_ms_per_step": uk_ms,
                     "note": "path is fp64-VALU bound, not HBM bound (SURVEY.md 8d); see fp64"},
        "fp64": {"achieved": ach_tflops, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach_tflops / FP64_VEC_PEAK_TFLOPS, "flops_per_solve": flops_per_solve,
                 "systems_by_matrix_rows": {str(int(e)): int(c) for e, c in zip(edges, rows_hist) if c}},
        "timing_ms": {k: float(np.mean([t[k] for t in kern])) for k in
                      ("tile_cand_ms", "select_ms", "uk_ms", "total_ms")},
    }
    if shared:
        res["note"] = "control-flow run: %d ranks share ONE GPU over gloo (fewer GPUs than ranks); not a scaling figure" % world

    # ---- daily record: the path that produces cell-DAYS (N = 1) --------------------------------------
    if world == 1 and not args.no_daily:
        import datetime as dt
        from topowx_amd.dates import get_days_metadata
        days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1980 + args.daily_years, 12, 31))
        nd = int(days.size)
        sn = synth.make_stations(base["bbox"], args.nstns, synth.CONFIGS["C2"][5], "tmin", days, with_obs=True)
        sx = synth.make_stations(base["bbox"], args.nstns, synth.CONFIGS["C2"][5], "tmax", days, with_obs=True)
        ctx.set_stations(_lib.TMIN, sn)
        ctx.set_stations(_lib.TMAX, sx)
        outs = {k: torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
                for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax")}
        d_dn = torch.full((nd, Y, X), int(_lib.FILL_I2), dtype=torch.int16, device=dev)
        d_dx = torch.full((nd, Y, X), int(_lib.FILL_I2), dtype=torch.int16, device=dev)
        o2 = _lib.TwxGridOut(outs["norm_tmin"].data_ptr(), outs["se_tmin"].data_ptr(), outs["norm_tmax"].data_ptr(),
                             outs["se_tmax"].data_ptr(), d_dn.data_ptr(), d_dx.data_ptr(), d_ninv.data_ptr(),
                             d_stat.data_ptr())
        both = _lib.VAR_TMIN_BIT | _lib.VAR_TMAX_BIT
        el2, k2 = timed(lambda: ctx.interp_grid_dev(g, o2, both, stream))
        ok2 = int((d_stat.cpu().numpy() == 0).sum())
        cell_days = ok2 * nd * 2
        tm = {k: float(np.mean([t[k] for t in k2])) for k in ("tile_cand_ms", "select_ms", "uk_ms", "gwr_ms",
                                                              "daily_ms", "fix_ms", "total_ms")}
        kan = ctx.last_bandwidths(_lib.TMIN).ravel()      # kriging bandwidths (GWR ones are of the same ladder)
        dgbs = ALG_BYTES_PER_CELL_DAY * cell_days / (tm["daily_ms"] * 1e-3) / 1e9
        res["daily"] = {
            "value": cell_days * args.steps / el2, "unit": "cell-days/s (whole path: selection + 24 normals per cell + GWR + "
                                                          "daily int16 + fixer, outputs resident in HBM)",
            "workload": "the same %dx%d tile and %d stations per variable, Tmin + Tmax, %d days (%d years), int16 daily "
                        "outputs + normals + SE + ninvalid" % (Y, X, args.nstns, nd, args.daily_years),
            "ms_per_step": el2 / args.steps * 1e3, "cell_days_per_step": cell_days, "cells_ok": ok2,
            "timing_ms": tm,
            "daily_kernel": {"kernel": "k_tile_union + k_daily_tile (+ k_row_offsets, k_daily_ok, k_daily_tile_gather)", "ms_per_step": tm["daily_ms"],
                             "cell_days_per_s": cell_days / (tm["daily_ms"] * 1e-3),
                             "roofline": {"bound": "hbm", "achieved": dgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": dgbs / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_launch": ALG_BYTES_PER_CELL_DAY * cell_days,
                                          "note": "every cell-day is a ~80-term dot product over observation rows; the rows "
                                                  "of a tile-month are staged in LDS (VALU / LDS bound), DESIGN.md section 4"}},
            "mean_nnghs": float(kan[kan > 0].mean()),
        }
        # packed int16 days against the oracle on a 4 x 4 window (integer output: identical except isolated +-1 LSB
        # where the fp64 value sits on a 0.005 rounding boundary and the summation order decides)
        if not args.no_cpu_baseline:
            from oracle import pyoracle as orc
            orc.build()
            r0w, c0w = min(100, Y - 4), min(60, X - 4)          # (inside the tile also for reduced --size runs)
            rs, cs = slice(r0w, r0w + 4), slice(c0w, c0w + 4)
            want = orc.interp_grid(orc.Db(sn), orc.Db(sx), orc.params(), grid, daily=True, nthreads=os.cpu_count() or 1,
                                   rows=rs, cols=cs)
            dd = np.concatenate([np.abs(d_dn[:, rs, cs].cpu().numpy().astype(np.int32) - want["daily_tmin"].astype(np.int32)).ravel(),
                                 np.abs(d_dx[:, rs, cs].cpu().numpy().astype(np.int32) - want["daily_tmax"].astype(np.int32)).ravel()])
            res["daily"]["packed_int16_vs_oracle"] = {"cells": 16, "values": int(dd.size), "identical_frac": float((dd == 0).mean()),
                                                      "max_abs_lsb": int(dd.max()),
                                                      "ninvalid_equal": bool(np.array_equal(d_ninv[rs, cs].cpu().numpy(), want["ninvalid"]))}
        del d_dn, d_dx, outs, sn, sx
        torch.cuda.empty_cache()
        # the same tile streamed: outputs of tile t travel to pinned host memory while tile t + 1 is computed
        # (twx_stream_*); end to end = host wall clock from the first submit to the last tile in host memory
        if args.stream_tiles > 0:
            ts = ctx.stream(Y, X, daily=True, nslots=2)
            ts.submit(0, grid); ts.wait(0)                       # warm-up: workspace, pinned pages
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dev_ms = 0.0
            for i in range(args.stream_tiles):
                ts.submit(i & 1, grid)
                if i:
                    dev_ms += ts.wait((i - 1) & 1)["device_ms"]
            last = ts.wait((args.stream_tiles - 1) & 1)
            dev_ms += last["device_ms"]
            wall = time.perf_counter() - t1
            okc = int((last["status"] == 0).sum())
            out_bytes = sum(v.nbytes for k, v in last.items() if hasattr(v, "nbytes"))
            ts.close()
            e2e = okc * nd * 2 * args.stream_tiles / wall
            res["daily"]["stream"] = {
                "tiles": args.stream_tiles, "end_to_end_cell_days_per_s": e2e,
                "device_only_cell_days_per_s": okc * nd * 2 * args.stream_tiles / (dev_ms * 1e-3),
                "ratio": e2e / (okc * nd * 2 * args.stream_tiles / (dev_ms * 1e-3)),
                "wall_s": wall, "device_ms_per_tile": dev_ms / args.stream_tiles,
                "d2h_bytes_per_tile": out_bytes, "d2h_GBps_if_exposed": out_bytes * args.stream_tiles / wall / 1e9,
                "note": "host pointers in, pinned host memory out (PCIe-inclusive; never the headline value)"}

    # ---- CPU baseline: the oracle on bounded samples of the headline workload ---------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        orc.build()
        cores = os.cpu_count() or 1
        db = orc.Db(stn)
        # all cores: >= 64 cells per thread so that the threads are loaded (not fork/join time)
        n_all = args.cpu_sample or int(min(Y, max(16, np.ceil(np.sqrt(64.0 * cores)))))
        t1 = time.perf_counter()
        ref = orc.interp_grid(db, None, orc.params(), grid, nthreads=cores, rows=slice(0, n_all), cols=slice(0, n_all))
        dt_all = time.perf_counter() - t1
        ok_all = int((ref["status"] == 0).sum())
        n_one = min(Y, 64)                        # ~10 s of one core
        t1 = time.perf_counter()
        ref1 = orc.interp_grid(db, None, orc.params(), grid, nthreads=1, rows=slice(0, n_one), cols=slice(0, n_one))
        dt_one = time.perf_counter() - t1
        ok_one = int((ref1["status"] == 0).sum())
        res["cpu_baseline"] = {
            "value": ok_all * 12 / dt_all, "unit": "cell-months/s", "cores": cores, "kind": "port",
            "sample": "%dx%d cell window of the same tile (%.0f cells per thread), all 12 months, OpenMP over cells "
                      "(%.1f s wall)" % (n_all, n_all, ok_all / cores, dt_all),
            "single_core": {"value": ok_one * 12 / dt_one, "unit": "cell-months/s", "cores": 1,
                            "sample": "%dx%d cell window, all 12 months (%.1f s)" % (n_one, n_one, dt_one)}}
        got = d_norm[:, :n_all, :n_all].cpu().numpy()
        res["parity_max_abs_degC"] = float(np.abs(got.astype(np.float64) - ref["norm_tmin"]).max())
        # the same kernels against the 40-digit arbiter of the kriging system (oracle/arbiter.py): two cells of the
        # tile, their own smoothed bandwidth and variogram
        try:
            from oracle import arbiter
            cdb = db.cols
            worst = 0.0
            for (r, q, m) in ((17 % Y, 200 % X, 1), (120 % Y, 40 % X, 7)):   # (inside the tile also for reduced --size runs)
                pts = ctx.make_pts(grid["lon"][q], grid["lat"][r], grid["elev"][r, q], grid["tdi"][r, q], grid["lst_night"][:, r, q])
                mean, var, used, st, ngh = ctx.krig_points(_lib.TMIN, pts, m, want_idx=True)
                pt = orc.make_pt(grid["lon"][q], grid["lat"][r], grid["elev"][r, q], grid["tdi"][r, q], grid["lst_night"][:, r, q])
                vp = np.zeros(3)
                import ctypes as C
                rc, idx, _, wgt = orc.select(db, grid["lat"][r], grid["lon"][q], int(used[0]))
                orc.lib().orc_smooth_vario(cdb["vario_nug"][m - 1].ctypes.data_as(C.POINTER(C.c_double)),
                                           cdb["vario_psill"][m - 1].ctypes.data_as(C.POINTER(C.c_double)),
                                           cdb["vario_rng"][m - 1].ctypes.data_as(C.POINTER(C.c_double)),
                                           idx.ctypes.data_as(C.POINTER(C.c_int32)), wgt.ctypes.data_as(C.POINTER(C.c_double)),
                                           C.c_int(int(used[0])), vp.ctypes.data_as(C.POINTER(C.c_double)))
                ix = ngh[0, :used[0]]
                am, av = arbiter.uk(cdb["lon"][ix], cdb["lat"][ix], cdb["elev"][ix], cdb["lst"][m - 1, ix], cdb["norm"][m - 1, ix],
                                    (grid["lon"][q], grid["lat"][r], float(grid["elev"][r, q]), float(grid["lst_night"][m - 1, r, q])),
                                    *vp)
                worst = max(worst, abs(mean[0] - am), abs(var[0] - av))
            res["parity_vs_arbiter_max_abs_degC"] = worst
        except ImportError:
            pass
    if rank == 0:
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
