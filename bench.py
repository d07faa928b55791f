#!/usr/bin/env python3
"""Benchmark of the interpolation hot path on MI355X.

One "step" = one pass of the hot path (twx_interp_grid_dev) over one synthetic
250x250 30-arcsec tile with ~10k stations, 12 monthly Tmin normals + standard
errors (BASELINE.json configs[1]); inputs are resident in HBM before the timed
region.  With N > 1 every rank interpolates its OWN tile against a replicated
station table (tiles partition embarrassingly): weak scaling, no
data-path collective.

Prints ONE JSON line on rank 0 (contract in the task description) carrying
``roofline`` (dominant kernel = k_uk, HIP-event timed inside the library on the
launch stream) and ``cpu_baseline`` (the CPU oracle on a bounded sample of the
same workload, all host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_CELL_MONTH = 13.1   # Tmin-only normals, 61 B in + 96 B out per cell / 12
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VEC_PEAK_TFLOPS = 78.6       # vendor fp64 vector peak


def uk_flops(k):
    """Algorithmic fp64 flops of one (cell, month) kriging system."""
    k = np.asarray(k, np.float64)
    p = 4
    return k ** 3 / 3.0 + (p + 3) * k ** 2 + 60.0 * k * (k - 1) / 2.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=250, help="tile edge in cells (default: the C2 tile)")
    ap.add_argument("--nstns", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=56, help="edge of the CPU-baseline sample window")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from topowx_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TWX_BENCH_BACKEND=gloo + TWX_BENCH_SHARE_GPU=1: control-flow check of the N > 1 path on a 1-GPU box
    backend = os.environ.get("TWX_BENCH_BACKEND", "nccl")
    if os.environ.get("TWX_BENCH_SHARE_GPU") == "1":
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
    d_in = {k: up(v) for k, v in a.items() if k != "lst_day"}
    d_norm = torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
    d_se = torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
    d_ninv = torch.full((Y, X), int(_lib.FILL_I4), dtype=torch.int32, device=dev)
    d_stat = torch.full((Y, X), -1, dtype=torch.int32, device=dev)
    g = _lib.TwxGrid(Y, X, d_in["mask"].data_ptr(), d_in["lat"].data_ptr(), d_in["lon"].data_ptr(),
                     d_in["elev"].data_ptr(), d_in["tdi"].data_ptr(), d_in["climdiv"].data_ptr(),
                     d_in["lst_night"].data_ptr(), None)
    o = _lib.TwxGridOut(d_norm.data_ptr(), d_se.data_ptr(), None, None, None, None, d_ninv.data_ptr(),
                        d_stat.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.interp_grid_dev(g, o, _lib.VAR_TMIN_BIT, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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

    status = d_stat.cpu().numpy()
    ncell_ok = int((status == 0).sum())
    units_per_step = ncell_ok * 12                     # (cell, month) outputs, each mean + SE
    value = world * units_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel (k_uk) --------------------------------------------
    uk_ms = float(np.mean([t["uk_ms"] for t in kern]))
    launches = max(1, int(kern[-1]["uk_launches"]))
    solves = int(kern[-1]["uk_solves"])
    ach_gbs = ALG_BYTES_PER_CELL_MONTH * solves / (uk_ms * 1e-3) / 1e9
    # bandwidths actually used by the timed steps (diagnostic accessor, no extra launches)
    ks = ctx.last_bandwidths(_lib.TMIN).ravel()
    flops_per_solve = float(uk_flops(ks[ks > 0]).mean())
    ach_tflops = flops_per_solve * solves / (uk_ms * 1e-3) / 1e12

    # HBM bytes per k_uk launch from the PMC passes of THIS workload (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs; summary committed under profiles/): PMC counters cannot be read
    # from inside the process, so the committed measurement is quoted when the workload matches.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r1_bench_hbm_traffic.json")
    if world == 1 and args.size == 250 and args.nstns == 10000 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = tj["FETCH_SIZE"]["k_uk_per_launch_bytes"] + tj["WRITE_SIZE"]["k_uk_per_launch_bytes"]
        traffic_src = "profiles/r1_bench_hbm_traffic.json (FETCH_SIZE + WRITE_SIZE, KB units x 1024, per launch; " \
                      "4/8-byte loads: the guide's x2 wide-read correction does not apply; almost all of it is the " \
                      "per-cell pair-distance cache that the cell's 12 monthly systems share, DESIGN.md section 4)"

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
                     "kernel": "universal kriging: k_cell_dist + k_ukw<..> + k_uk<..> (%d launches per step)" % launches,
                     "kernel_ms_per_step": uk_ms,
                     "note": "path is fp64-VALU bound, not HBM bound (SURVEY.md 8d); see fp64"},
        "fp64": {"achieved": ach_tflops, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach_tflops / FP64_VEC_PEAK_TFLOPS, "flops_per_solve": flops_per_solve},
        "timing_ms": {k: float(np.mean([t[k] for t in kern])) for k in
                      ("tile_cand_ms", "select_ms", "uk_ms", "total_ms")},
    }

    # ---- CPU baseline: the oracle on a bounded sample of the same workload ---------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        orc.build()
        n = min(args.cpu_sample, Y)
        cores = os.cpu_count() or 1
        db = orc.Db(stn)
        t1 = time.perf_counter()
        ref = orc.interp_grid(db, None, orc.params(), grid, nthreads=cores, rows=slice(0, n), cols=slice(0, n))
        dt = time.perf_counter() - t1
        okc = int((ref["status"] == 0).sum())
        res["cpu_baseline"] = {"value": okc * 12 / dt, "unit": "cell-months/s", "cores": cores, "kind": "port",
                               "sample": "%dx%d cell window of the same tile, all 12 months, OpenMP over cells "
                                         "(%.1f s wall)" % (n, n, dt)}
        got = d_norm[:, :n, :n].cpu().numpy()
        res["parity_max_abs_degC"] = float(np.abs(got.astype(np.float64) - ref["norm_tmin"]).max())
    if rank == 0:
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
