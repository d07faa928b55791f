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
                launch stream; HBM as BASELINE.json asks, the binding fp64-vector figure next to it (``fp64``:
                nominal = SURVEY 8d's flops, executed = factorisation + border only)
  daily         (N = 1) the cell-DAY producing path, timed the same way: the same tile, Tmin + Tmax, normals +
                GWR + 10 years of daily int16 values + Tmin>=Tmax fixer, outputs resident in HBM
  configs       (N = 1) the other BASELINE.json configurations, driver-timed in the same run: ``c4_tile`` (the
                25 203-day tile of configs[3]), ``c5`` (configs[4]: step21 -> 22 -> 23 -> 24 over all stations of the
                12 000-station seed-2 database), ``c3`` (configs[2]: the FULL 3250x7000 masked grid, Tmin + Tmax
                normals, through topowx_amd.driver on one GPU)
  strong        (N > 1) the tile farm of topowx_amd.driver on ONE fixed masked grid (a 750x7000 strip of configs[2]'s
                grid): LPT tile deal, per-rank device-resident tiles, RCCL gather of the normals mosaic on device
                tensors; ``--scaling strong`` makes it the top-level line.  ``strong.daily``: the same deal on the
                daily (streamed) path -- every rank streams its tiles' int16 days to pinned host memory, no gather
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
ALG_BYTES_PER_CELL_MONTH_2V = 12.7   # Tmin + Tmax normals: 305 B / 24 cell-months
ALG_BYTES_PER_CELL_DAY = 2.03     # int16 out + amortised inputs / observation matrix
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VEC_PEAK_TFLOPS = 78.6       # vendor fp64 vector peak
DTYPE = "f64 (f32 pair distances)"
DTYPE_NOTE = ("everything that accumulates -- selection, Cholesky, solves, GWR, daily sums -- is fp64; the off-diagonal covariance "
              "entries of well-conditioned kriging systems come from an fp32 pair-distance cache and v_exp_f32 (~2e-7 psill "
              "per entry); systems that would amplify that beyond 1e-5 degC (amplification > 8, twx_select.h: uk_needs_f64) "
              "are built in fp64 (k_uk<NB, 2, 1>)")


def uk_flops(k):
    """Algorithmic fp64 flops of one (cell, month) kriging system: Cholesky + border rows +
    pair distances / covariances evaluated per system."""
    k = np.asarray(k, np.float64)
    return uk_flops_executed(k) + uk_flops_distance(k)


def uk_flops_executed(k):
    """What the kriging kernels execute per system today: k^3/3 (Cholesky) + 7 k^2 (the seven border rows); the pair
    distances are evaluated once per tile and station pair (k_tile_dist), not per system."""
    k = np.asarray(k, np.float64)
    return k ** 3 / 3.0 + 7.0 * k ** 2


def uk_flops_distance(k):
    """SURVEY 8d's nominal distance / covariance term: ~60 flops for each of the k (k - 1) / 2 pairs of a system."""
    k = np.asarray(k, np.float64)
    return 60.0 * k * (k - 1) / 2.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=250, help="tile edge in cells (default: the C2 tile)")
    ap.add_argument("--nstns", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-daily", action="store_true", help="skip the daily (cell-days) record")
    ap.add_argument("--daily-years", type=int, default=10, help="years of days of the daily record (from --daily-year0)")
    ap.add_argument("--daily-year0", type=int, default=1981, help="first year of the daily record's axis (configs[3]'s own axis: 1948 with --daily-years 69)")
    ap.add_argument("--stream-tiles", type=int, default=4, help="tiles of the streamed (PCIe-inclusive) daily record; 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=0, help="edge of the all-core CPU sample window (0 = auto)")
    ap.add_argument("--int16-window", type=int, default=24, help="edge of the cell window whose packed int16 days are compared with the oracle")
    ap.add_argument("--no-configs", action="store_true", help="skip the c4_tile / c5 / c3 records (N = 1)")
    ap.add_argument("--configs", default="c2_fitted,c4_tile,c5,c3,c3_fitted,c4",
                    help="which of the other configurations to time (c2_fitted = the headline tile under the variograms step21 -> "
                         "step22 fit on its own database; c3 = the full configs[2] grid; c3_fitted = the same under the variograms step21 -> step22 fit on its 12 000-station tables; c4 = configs[3] itself: the full grid x "
                         "25 203 days streamed to pinned host memory; c3_strip = the 750x7000 strip of the strong record)")
    ap.add_argument("--c4-tiles", type=int, default=0, help="c4 record: only the first N tiles of the deal (0 = all 323; tests)")
    ap.add_argument("--c4-rows", type=int, default=0, help="c4 record on a cut of the grid (tests; 0 = the full 3250x7000 grid)")
    ap.add_argument("--c4-cols", type=int, default=0)
    ap.add_argument("--c4-years", type=int, default=69, help="c4 record: years of days from 1948 (69 = 1948-2016, 25 203 days)")
    ap.add_argument("--c4-precision", default="auto", choices=("auto", "fast", "exact"),
                    help="c4 record: covariance build of the streamed run (driver.PrecisionPolicy; auto = exact while it is free)")
    ap.add_argument("--no-c4-deflate", action="store_true", help="c4 record: skip the second pass with the daily values deflated on the GPU")
    ap.add_argument("--c4-sink-tiles", type=int, default=16, help="c4 record: tiles written into NetCDF-4 tile files by ncio.TileSink (0 = skip)")
    ap.add_argument("--c4-sink-dir", default=None, help="where (default: /dev/shm when it has 20 GB free, else $TMPDIR)")
    ap.add_argument("--c4-sink-threads", type=int, default=0, help="TileSink workers (0 = min(32, cpu count))")
    ap.add_argument("--force-configs", action="store_true", help="time them also on a reduced --size (tests)")
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto: N = 1 headline (+ configs); N > 1 weak headline + a 'strong' record.  strong: the tile farm "
                         "of topowx_amd.driver on one fixed masked grid is the top-level line")
    ap.add_argument("--strip-rows", type=int, default=750)
    ap.add_argument("--strip-cols", type=int, default=7000)
    ap.add_argument("--strip-nstns", type=int, default=12000)
    ap.add_argument("--strip-tile", type=int, default=250)
    ap.add_argument("--strong-steps", type=int, default=3, help="timed passes of the strong record when it is not the top-level line")
    ap.add_argument("--c5-years", type=int, default=3)
    ap.add_argument("--c5-nstns", type=int, default=12000, help="stations of the configs[4] database (SURVEY 8d: 12 000, seed 2)")
    ap.add_argument("--c3-steps", type=int, default=2, help="timed passes over the full configs[2] grid")
    ap.add_argument("--strong-daily-rows", type=int, default=500, help="rows of the strong daily record's grid (0 = skip)")
    ap.add_argument("--strong-daily-cols", type=int, default=1000)
    ap.add_argument("--strong-daily-years", type=int, default=3)
    ap.add_argument("--dump-mosaic", default=None, help="strong mode: rank 0 writes the gathered mosaic here (.npz)")
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
    """HBM bytes from the newest committed PMC reduction (profiles/r*_bench_hbm_traffic.json): per kriging launch
    (roofline.traffic) and per daily step (daily.traffic)."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_hbm_traffic.json")))
    if not paths:
        return None, None, None
    tj = json.load(open(paths[-1]))
    traffic = tj["FETCH_SIZE"]["k_uk_per_launch_bytes"] + tj["WRITE_SIZE"]["k_uk_per_launch_bytes"]
    daily = None
    if "daily_path_per_step_bytes" in tj["FETCH_SIZE"]:
        daily = tj["FETCH_SIZE"]["daily_path_per_step_bytes"] + tj["WRITE_SIZE"]["daily_path_per_step_bytes"]
    src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this workload, KB units x "
           "1024; 4-byte loads of the fp32 pair-distance cache: the guide's x2 correction for "
           "16-B/lane streaming reads does not apply)" % os.path.basename(paths[-1]))
    # do the committed counters belong to the kernels in the tree?  (tests/test_profiles_fresh.py fails when they do not)
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    try:
        import kernel_hash
        if tj.get("kernel_sources_sha16") != kernel_hash.kernel_sources_sha16():
            src += " -- STALE: collected on other kernel sources than this tree's (re-run tests/tools/collect_round.sh)"
    except ImportError:
        pass
    return traffic, daily, src


class Env(object):
    """Process-group / device context of this rank."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # TWX_BENCH_BACKEND=gloo + TWX_BENCH_SHARE_GPU=1: control-flow check of the N > 1 path on a 1-GPU box
        self.backend = os.environ.get("TWX_BENCH_BACKEND", "nccl")
        self.shared = os.environ.get("TWX_BENCH_SHARE_GPU") == "1"
        if self.shared:
            self.local = 0
        # TWX_BENCH_FORCE_PG=1: a process group -- and every collective of the N > 1 path -- also at world 1 (a one-rank RCCL
        # communicator on the 1-GPU box: tests/test_gpu_rccl_smoke.py)
        self.pg = self.world > 1 or os.environ.get("TWX_BENCH_FORCE_PG") == "1"
        if self.pg and self.world == 1:
            os.environ.setdefault("MASTER_PORT", "29731")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if self.pg:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group(self.backend)
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)

    def barrier(self):
        if self.pg:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.pg:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_scalar(self, x):
        if not self.pg:
            return [float(x)]
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        parts = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]

    def sum_over_ranks(self, x):
        return float(sum(self.all_gather_scalar(x)))


def timed(env, step, steps, warmup, after_step=None):
    """W untimed + K timed steps between barrier + synchronize; max over ranks."""
    for _ in range(warmup):
        step()
        if after_step:
            after_step(False)
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if after_step:
            after_step(True)
    env.barrier()
    return env.max_over_ranks(time.perf_counter() - t0)


# =====================================================================================================================
# daily record (default: the C2 tile with 10 years of days; c4_tile: 1948-2016)
# =====================================================================================================================
def daily_record(env, args, base, grid, g, d_ninv, d_stat, day0, day1, label, steps, warmup, stream_tiles, int16_window,
                 traffic_bytes=None, traffic_src=None):
    """Tmin + Tmax normals + daily int16 + fixer of the tile on the day axis day0 .. day1, in a context of its own (a
    context's day axis is fixed once observations are loaded)."""
    torch = env.torch
    from topowx_amd import _lib, synth
    from topowx_amd.dates import get_days_metadata
    dev = env.dev
    Y = X = args.size
    days = get_days_metadata(day0, day1)
    nd = int(days.size)
    t_s = time.perf_counter()
    sn = synth.make_stations(base["bbox"], args.nstns, synth.CONFIGS["C2"][5], "tmin", days, with_obs=True)
    sx = synth.make_stations(base["bbox"], args.nstns, synth.CONFIGS["C2"][5], "tmax", days, with_obs=True)
    ctx = _lib.Context(device=env.local)
    ctx.set_stations(_lib.TMIN, sn)
    ctx.set_stations(_lib.TMAX, sx)
    setup_s = time.perf_counter() - t_s
    outs = {k: torch.full((12, Y, X), float(_lib.FILL_F4), dtype=torch.float32, device=dev)
            for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax")}
    d_dn = torch.full((nd, Y, X), int(_lib.FILL_I2), dtype=torch.int16, device=dev)
    d_dx = torch.full((nd, Y, X), int(_lib.FILL_I2), dtype=torch.int16, device=dev)
    o2 = _lib.TwxGridOut(outs["norm_tmin"].data_ptr(), outs["se_tmin"].data_ptr(), outs["norm_tmax"].data_ptr(),
                         outs["se_tmax"].data_ptr(), d_dn.data_ptr(), d_dx.data_ptr(), d_ninv.data_ptr(),
                         d_stat.data_ptr())
    both = _lib.VAR_TMIN_BIT | _lib.VAR_TMAX_BIT
    stream = torch.cuda.current_stream().cuda_stream
    k2 = []
    el2 = timed(env, lambda: ctx.interp_grid_dev(g, o2, both, stream), steps, warmup,
                lambda keep: k2.append(ctx.timing()) if keep else ctx.timing())
    ok2 = int((d_stat.cpu().numpy() == 0).sum())
    cell_days = ok2 * nd * 2
    tm = {k: float(np.mean([t[k] for t in k2])) for k in ("tile_cand_ms", "select_ms", "uk_ms", "gwr_ms",
                                                          "daily_ms", "fix_ms", "total_ms")}
    kan = ctx.last_bandwidths(_lib.TMIN).ravel()      # kriging bandwidths (GWR ones are of the same ladder)
    dgbs = ALG_BYTES_PER_CELL_DAY * cell_days / (tm["daily_ms"] * 1e-3) / 1e9
    ninv = d_ninv.cpu().numpy()
    rec = {
        "value": cell_days * steps / el2, "unit": "cell-days/s (whole path: selection + 24 normals per cell + GWR + "
                                                 "daily int16 + fixer, outputs resident in HBM)",
        "workload": "%s: the %dx%d tile and %d stations per variable, Tmin + Tmax, %d days (%s .. %s), int16 daily "
                    "outputs + normals + SE + ninvalid" % (label, Y, X, args.nstns, nd, day0.isoformat(), day1.isoformat()),
        "steps": steps, "warmup": warmup,
        "ms_per_step": el2 / steps * 1e3, "cell_days_per_step": cell_days, "cells_ok": ok2,
        "cells_with_fixed_days": int((ninv[d_stat.cpu().numpy() == 0] > 0).sum()),
        "timing_ms": tm, "setup_s": setup_s,
        "daily_kernel": {"kernel": "k_daily_tile (+ k_row_offsets, k_daily_ok, k_daily_tile_gather)", "ms_per_step": tm["daily_ms"],
                         "cell_days_per_s": cell_days / (tm["daily_ms"] * 1e-3),
                         "roofline": {"bound": "hbm", "achieved": dgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": dgbs / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_launch": ALG_BYTES_PER_CELL_DAY * cell_days,
                                      "note": "every cell-day is a ~80-term dot product over observation rows; the rows "
                                              "of a tile-month are staged in LDS (VALU / LDS bound), DESIGN.md section 4"}},
        "mean_nnghs": float(kan[kan > 0].mean()),
    }
    if traffic_bytes is not None:
        # measured HBM bytes of the launches behind daily_ms + gwr_ms (k_perm, k_tile_uidx, k_gwr_z_cell, k_daily_tile, ...) per step
        alg = ALG_BYTES_PER_CELL_DAY * cell_days
        rec["traffic"] = {"bytes_per_step": traffic_bytes, "algorithmic_bytes_per_step": alg, "ratio": traffic_bytes / alg,
                          "measured_in_this_run": False, "source": traffic_src}
    # packed int16 days against the oracle on a window of cells (integer output: identical except isolated +-1 LSB
    # where the fp64 value sits on a 0.005 rounding boundary and the summation order decides)
    if not args.no_cpu_baseline and int16_window > 0:
        from oracle import pyoracle as orc
        orc.build()
        w = min(int16_window, Y, X)
        r0w, c0w = min(100, Y - w), min(60, X - w)          # (inside the tile also for reduced --size runs)
        rs, cs = slice(r0w, r0w + w), slice(c0w, c0w + w)
        t1 = time.perf_counter()
        want = orc.interp_grid(orc.Db(sn), orc.Db(sx), orc.params(), grid, daily=True, nthreads=os.cpu_count() or 1,
                               rows=rs, cols=cs)
        dd = np.concatenate([np.abs(d_dn[:, rs, cs].cpu().numpy().astype(np.int32) - want["daily_tmin"].astype(np.int32)).ravel(),
                             np.abs(d_dx[:, rs, cs].cpu().numpy().astype(np.int32) - want["daily_tmax"].astype(np.int32)).ravel()])
        nerr = float(max(np.abs(outs[k][:, rs, cs].cpu().numpy().astype(np.float64) - want[k]).max()
                         for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax")))
        rec["packed_int16_vs_oracle"] = {"cells": w * w, "values": int(dd.size), "identical_frac": float((dd == 0).mean()),
                                         "flips": int((dd != 0).sum()), "flip_rate": float((dd != 0).mean()),
                                         "max_abs_lsb": int(dd.max()),
                                         "ninvalid_equal": bool(np.array_equal(ninv[rs, cs], want["ninvalid"])),
                                         "status_equal": bool(np.array_equal(d_stat[rs, cs].cpu().numpy(), want["status"])),
                                         "normals_max_abs_degC": nerr, "oracle_s": time.perf_counter() - t1}
    del d_dn, d_dx, outs, sn, sx
    torch.cuda.empty_cache()
    # the same tile streamed: outputs of tile t travel to pinned host memory while tile t + 1 is computed
    # (twx_stream_*); end to end = host wall clock from the first submit to the last tile in host memory
    if stream_tiles > 0:
        ts = ctx.stream(Y, X, daily=True, nslots=2)
        ts.submit(0, grid); ts.wait(0)                       # warm-up: workspace, pinned pages
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dev_ms = 0.0
        for i in range(stream_tiles):
            ts.submit(i & 1, grid)
            if i:
                dev_ms += ts.wait((i - 1) & 1)["device_ms"]
        last = ts.wait((stream_tiles - 1) & 1)
        dev_ms += last["device_ms"]
        wall = time.perf_counter() - t1
        okc = int((last["status"] == 0).sum())
        out_bytes = sum(v.nbytes for k, v in last.items() if hasattr(v, "nbytes"))
        ts.close()
        e2e = okc * nd * 2 * stream_tiles / wall
        rec["stream"] = {
            "tiles": stream_tiles, "end_to_end_cell_days_per_s": e2e,
            "device_only_cell_days_per_s": okc * nd * 2 * stream_tiles / (dev_ms * 1e-3),
            "ratio": e2e / (okc * nd * 2 * stream_tiles / (dev_ms * 1e-3)),
            "wall_s": wall, "device_ms_per_tile": dev_ms / stream_tiles,
            "d2h_bytes_per_tile": out_bytes, "d2h_GBps_if_exposed": out_bytes * stream_tiles / wall / 1e9,
            "note": "host pointers in, pinned host memory out (PCIe-inclusive; never the headline value)"}
    ctx.close()
    return rec


def shared_stations(env, key, build):
    """A synthetic station database every rank needs: rank 0 builds it ONCE and saves it (uncompressed .npz in /dev/shm),
    the others load it after a barrier -- N-rank setup is one build + N loads, not N builds on the same host cores."""
    from topowx_amd import stationdb as sdb
    if env.world == 1:
        return build()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(base, "twx_bench_%s_%s.npz" % (os.environ.get("MASTER_PORT", "0"), key))
    db = None
    if env.rank == 0:
        db = build()
        db.save(path, compress=False)
    env.dist.barrier()
    if env.rank != 0:
        db = sdb.StationDataWrkChk.load(path)
    env.dist.barrier()
    if env.rank == 0:
        os.remove(path)
    return db


# =====================================================================================================================
# the tile farm of topowx_amd.driver on ONE fixed masked grid (strong scaling; at N = 1 the c3 / c3_strip records)
# =====================================================================================================================
def strip_run(env, args, steps, warmup, spot_check, full=False, fitted=False):
    """BASELINE.json configs[2] shape on a strip of the seed-7 masked CONUS-shaped grid: Tmin + Tmax normals + SE of
    every valid cell, tiles dealt to the ranks with driver.assign_tiles (LPT), every rank's tiles computed device-
    resident into the send buffer of ONE dist.gather (RCCL over xGMI) that assembles the four mosaics on rank 0.
    One step = the whole strip: deal + all tiles + gather.  Reference shape: step25:266-314 (coordinator farm)."""
    torch = env.torch
    from topowx_amd import _lib, driver, synth
    T = args.strip_tile
    t_s = time.perf_counter()
    if full:       # BASELINE.json configs[2] itself: the whole 3250x7000 grid (fits one GPU: ~13 GB of HBM)
        grid = synth.make_grid("C3")
    else:          # rows of configs[2]'s grid south of 45 N (the C3 grid starts at 51.6 N): same generator, same mask seed
        grid = synth.make_grid("C3", nrows=args.strip_rows, ncols=args.strip_cols, lat_north=45.0, full_mask=False)
    tmin = shared_stations(env, "strip_tmin", lambda: synth.make_stations(grid["bbox"], args.strip_nstns, synth.CONFIGS["C3"][5], "tmin"))
    tmax = shared_stations(env, "strip_tmax", lambda: synth.make_stations(grid["bbox"], args.strip_nstns, synth.CONFIGS["C3"][5], "tmax"))
    fit_info = None
    if fitted:   # the same grid and bandwidths under the variograms step21 -> step22 fit on these very tables (6 stations per deg^2)
        tmin, fi_n = fit_table_variograms(env, tmin, "tmin")
        tmax, fi_x = fit_table_variograms(env, tmax, "tmax")
        fit_info = {"tmin": fi_n, "tmax": fi_x}
    ctx = _lib.Context(device=env.local)
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    tiles = driver.tile_list(grid["mask"], T, T)
    assignment = driver.assign_tiles(tiles, env.world)
    mine = assignment[env.rank]
    nmax = max(len(a) for a in assignment)
    dgrid = driver.upload_grid(grid, env.dev, mine, T, T)        # this rank's tiles only (1 / world of the grid per GPU)
    setup_s = time.perf_counter() - t_s
    shape = grid["mask"].shape
    state = {}
    dev_ms, gather_ms, tile_wall = [], [], []

    ukstats = {}

    def step():
        t0 = time.perf_counter()
        buf, stat, ms = driver.interp_tiles_device(ctx, dgrid, mine, T, T, nslots=nmax)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if env.pg:
            env.dist.barrier()          # so that gather_ms is the collective, not the wait for the slowest rank
        t2 = time.perf_counter()
        mosaic = driver.gather_mosaic_device(buf, assignment, shape, T, T, env.rank, env.world, backend=env.backend, collective=env.pg)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        state.update(buf=buf, stat=stat, mosaic=mosaic, ms=float(sum(ms)), tile_wall=t1 - t0, gather=(t3 - t2) * 1e3)

    def after(keep):
        if keep:
            dev_ms.append(state["ms"]); gather_ms.append(state["gather"]); tile_wall.append(state["tile_wall"])

    elapsed = timed(env, step, steps, warmup, after)
    if fitted:      # the launch statistics in a pass of their own: reading them waits for every tile (round 5 paid that inside the timing)
        driver.interp_tiles_device(ctx, dgrid, mine, T, T, nslots=nmax, stats=ukstats)
    cells_ok = int(env.sum_over_ranks(int((state["stat"][:len(mine)] == 0).sum().item()) if mine else 0))
    per_rank_ms = env.all_gather_scalar(float(np.mean(dev_ms)))
    per_rank_wall = env.all_gather_scalar(float(np.mean(tile_wall)) * 1e3)
    per_rank_cells = [sum(t[3] for t in a) for a in assignment]
    units = cells_ok * 24
    rec = {
        "value": units * steps / elapsed, "unit": "cell-months/s (Tmin + Tmax normals, mean + SE each)",
        "workload": ("c3: BASELINE.json configs[2] -- the FULL %dx%d seed-7 masked CONUS-shaped 30-arcsec grid, " % shape if full else
                     "c3_strip: %dx%d cells of the seed-7 masked CONUS-shaped 30-arcsec grid (rows south of 45 N of "
                     "BASELINE.json configs[2]), " % shape) +
                    "%d synthetic stations per variable, 12 monthly Tmin + Tmax normals + SE; %dx%d tiles dealt by "
                    "topowx_amd.driver.assign_tiles" % (args.strip_nstns, T, T),
        "n_gpus": env.world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "cells_valid": int((grid["mask"] != 0).sum()), "cells_ok": cells_ok, "tiles": len(tiles),
        "tiles_per_rank": [len(a) for a in assignment], "valid_cells_per_rank": per_rank_cells,
        "device_ms_per_rank": per_rank_ms, "tile_loop_wall_ms_per_rank": per_rank_wall,
        "imbalance_max_over_mean": max(per_rank_ms) / max(1e-9, float(np.mean(per_rank_ms))),
        "gather_ms": float(np.mean(gather_ms)),
        "gather_bytes_per_rank": int(nmax * 4 * 12 * T * T * 4),
        "gather": "one dist.gather of the [tiles, 4, 12, %d, %d] f4 device tensor per rank (%s)" % (
            T, T, "RCCL over xGMI" if env.backend == "nccl" and env.world > 1 else
            ("gloo through host memory: control-flow run" if env.world > 1 else
             ("single rank through a ONE-RANK RCCL communicator (TWX_BENCH_FORCE_PG=1)" if env.pg and env.backend == "nccl" else
              "single rank: device copies only"))),
        "process_group": ("%s, world %d" % (env.backend, env.world)) if env.pg else None,
        "hbm": {"algorithmic_bytes_per_cell_month": ALG_BYTES_PER_CELL_MONTH_2V,
                "achieved_GBps": ALG_BYTES_PER_CELL_MONTH_2V * units * steps / elapsed / 1e9,
                "frac_of_peak": ALG_BYTES_PER_CELL_MONTH_2V * units * steps / elapsed / 1e9 / (HBM_PEAK_GBS * env.world)},
        "setup_s": setup_s,
    }
    if fitted:
        rec["workload"] = rec["workload"].replace("c3: ", "c3_fitted: ", 1) + (
            "; vario_nug / vario_psill / vario_rng of both tables replaced by what step21 -> set_optim_nstns_tair_norm -> step22 fit on them")
        rec["fitted_variograms"] = fit_info
        rec["uk_solves"] = int(ukstats.get("uk_solves", 0))
        rec["systems_on_fp64_covariance_build"] = int(ukstats.get("uk_f64_solves", 0))
        rec["frac_on_fp64_covariance_build"] = rec["systems_on_fp64_covariance_build"] / max(1, rec["uk_solves"])
        rec["note"] = "systems counted in an untimed pass of their own; the timed pass is the code path of configs.c3"
    if env.rank == 0 and args.dump_mosaic:
        np.savez(args.dump_mosaic, **{k: v.cpu().numpy() for k, v in state["mosaic"].items()})
    if env.rank == 0 and spot_check and not args.no_cpu_baseline:
        # a 6x6 window of valid cells of the mosaic against the oracle
        from oracle import pyoracle as orc
        orc.build()
        m = grid["mask"] != 0
        rr, cc = np.nonzero(m[:-6, :-6] & m[6:, 6:] & m[:-6, 6:] & m[6:, :-6])
        worst, stat_eq = 0.0, True
        if rr.size:
            r0, c0 = int(rr[rr.size // 2]), int(cc[rr.size // 2])
            rs, cs = slice(r0, r0 + 6), slice(c0, c0 + 6)
            want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), grid, nthreads=min(8, os.cpu_count() or 1), rows=rs, cols=cs)
            okw = want["status"] == 0
            for k in driver.NORMAL_KEYS:
                got = state["mosaic"][k][:, rs, cs].cpu().numpy().astype(np.float64)
                worst = max(worst, float(np.abs(got - want[k])[:, okw].max()) if okw.any() else 0.0)
                stat_eq = stat_eq and bool(np.all(got[:, ~okw] == float(_lib.FILL_F4)))
            rec["spot_check_vs_oracle"] = {"cells": 36, "window": [r0, c0], "max_abs_degC": worst, "fills_where_oracle_fails": stat_eq}
    ctx.close()
    return rec


def strong_daily_run(env, args):
    """The daily (streamed) path under the same tile deal: every rank pushes its tiles through a TileStream
    (twx_stream_*: kernels of tile t + 1 overlap the copy-out of tile t) into pinned host memory; NO gather -- the daily
    int16 output (1.3 TB for configs[3]) is never assembled on a device: each rank's writer owns
    its tiles, as the reference's workers write their own chunks (step25:177-185).  One pass = every tile of a fixed
    masked grid; value = cell-days of all ranks / the slowest rank's wall."""
    import datetime as dt
    from topowx_amd import _lib, driver, synth
    from topowx_amd.dates import get_days_metadata
    T = args.strip_tile
    R, Cc = args.strong_daily_rows // T * T, args.strong_daily_cols // T * T
    grid = synth.make_grid("C3", nrows=R, ncols=Cc, lat_north=45.0, full_mask=False)
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1980 + args.strong_daily_years, 12, 31))
    nd = int(days.size)
    seed = synth.CONFIGS["C3"][5]
    t_s = time.perf_counter()
    tmin = shared_stations(env, "sd_tmin", lambda: synth.make_stations(grid["bbox"], args.strip_nstns, seed, "tmin", days, with_obs=True))
    tmax = shared_stations(env, "sd_tmax", lambda: synth.make_stations(grid["bbox"], args.strip_nstns, seed, "tmax", days, with_obs=True))
    ctx = _lib.Context(device=env.local)
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    setup_s = time.perf_counter() - t_s
    tiles = driver.tile_list(grid["mask"], T, T)
    assignment = driver.assign_tiles(tiles, env.world)
    mine = assignment[env.rank]
    acc = {"bytes": 0, "ok": 0}

    def sink(k, arrays):
        acc["bytes"] += sum(v.nbytes for v in arrays.values() if hasattr(v, "nbytes"))
        acc["ok"] += int((arrays["status"] == 0).sum())

    driver.interp_tiles_streamed(ctx, grid, mine[:1], T, T, daily=True, sink=lambda k, a: None)     # warm-up: workspace, pinned slots
    env.barrier()
    t0 = time.perf_counter()
    _, secs, dev_ms = driver.interp_tiles_streamed(ctx, grid, mine, T, T, daily=True, sink=sink) if mine else (None, 0.0, 0.0)
    wall_local = time.perf_counter() - t0
    env.barrier()
    wall = env.max_over_ranks(wall_local)
    ok = env.sum_over_ranks(acc["ok"])
    rec = {"value": ok * nd * 2 / wall, "unit": "cell-days/s (Tmin + Tmax daily int16 + normals + SE, streamed to pinned host memory; PCIe-inclusive)",
           "workload": "%dx%d cells of the seed-7 masked grid, %d stations per variable, %d days, %dx%d tiles dealt by assign_tiles; "
                       "every rank streams its own tiles (interp_tiles_streamed), no gather" % (R, Cc, args.strip_nstns, nd, T, T),
           "n_gpus": env.world, "scaling": "strong", "wall_s": wall, "cells_ok": int(ok), "days": nd,
           "tiles_per_rank": [len(a) for a in assignment],
           "wall_s_per_rank": env.all_gather_scalar(wall_local), "device_ms_per_rank": env.all_gather_scalar(dev_ms),
           "d2h_bytes_per_rank": [int(b) for b in env.all_gather_scalar(acc["bytes"])], "setup_s": setup_s}
    ctx.close()
    return rec


def fit_table_variograms(env, stn, var):
    """step21 -> set_optim_nstns_tair_norm -> step22 on a copy of a station table; returns a copy of ``stn`` whose vario_* columns
    are the fitted ones (bandwidth columns untouched, so that the systems are those of the un-fitted run) + the fit statistics."""
    from topowx_amd import stationdb as sdb, xval
    t0 = time.perf_counter()
    work = sdb.StationDataWrkChk(stn.stns.copy(), var, stn.days, None)
    ids = xval.xval_station_ids(work)
    kw = dict(stn_ids=ids, device=env.local)
    _, mae = xval.optim_nstns_norms(work, var, **kw)
    xval.set_optim_nstns_tair_norm(work, ids, mae)
    _, nug, psill, rng = xval.set_stn_variograms(work, var, **kw)
    fitted = sdb.StationDataWrkChk(stn.stns.copy(), var, stn.days, None)
    for m in range(1, 13):
        for par in (sdb.VARIO_NUG, sdb.VARIO_PSILL, sdb.VARIO_RNG):
            name = sdb.get_krigparam_varname(m, par)
            fitted.stns[name] = work.stns[name]
    fin = np.isfinite(nug) & np.isfinite(psill)
    ratio = nug[fin] / np.where(psill[fin] > 0, psill[fin], np.nan)
    ratio = ratio[np.isfinite(ratio)]
    okr = np.isfinite(rng) & (rng > 0)
    info = {"stations_fitted": int(len(ids)), "station_months_fitted": int(np.isfinite(nug).sum()), "fit_s": time.perf_counter() - t0,
            "nug_over_psill_quantiles_5_25_50_75_95": [float(x) for x in np.quantile(ratio, [.05, .25, .5, .75, .95])] if ratio.size else None,
            "frac_below_one_sixteenth": float((ratio < 1 / 16.).mean()) if ratio.size else None,
            "pure_nugget_frac": float((rng[np.isfinite(rng)] == 0).mean()) if np.isfinite(rng).any() else None,
            "range_km_quantiles_5_50_95": [float(x) for x in np.quantile(rng[okr], [.05, .5, .95])] if okr.any() else None}
    return fitted, info


def c2_fitted_record(env, args, ctx, stn, grid, g, o, d_norm, d_stat, stream, headline_uk_ms, headline_value):
    """The headline tile under the pipeline's OWN variograms: step21 (leave-one-out bandwidth
    optimisation, step21:34-64) -> ``set_optim_nstns_tair_norm`` -> step22 (every station's variogram fitted with the
    optimised bandwidths: nugget = min gamma, interp.R:304-359; step22:33-66) run on the C2 station database itself;
    then the SAME tile, bandwidths and launches as the headline, with only the ``vario_*`` columns replaced by the fitted
    ones -- so that ``uk_ms`` differs from the headline's by the systems routed to the fp64 covariance build alone."""
    from topowx_amd import _lib
    fitted, fi = fit_table_variograms(env, stn, "tmin")
    ctx.set_stations(_lib.TMIN, fitted, with_obs=False)
    kern = []
    steps = max(3, args.steps // 2)
    elapsed = timed(env, lambda: ctx.interp_grid_dev(g, o, _lib.VAR_TMIN_BIT, stream), steps, 1,
                    lambda keep: kern.append(ctx.timing()) if keep else ctx.timing())
    status = d_stat.cpu().numpy()
    ok = int((status == 0).sum())
    uk_ms = float(np.mean([t["uk_ms"] for t in kern]))
    rec = {"value": ok * 12 * steps / elapsed, "unit": "cell-months/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "workload": "c2_fitted: the headline's C2 tile, bandwidths and launches; vario_nug / vario_psill / vario_rng of all "
                       "%d cross-validated stations replaced by what step21 -> set_optim_nstns_tair_norm -> step22 fit on the same "
                       "database (topowx_amd.xval; %d station-months fitted)" % (fi["stations_fitted"], fi["station_months_fitted"]),
           "cells_ok": ok, "cells_failed_by_status": {str(int(c)): int(n) for c, n in zip(*np.unique(status[status != 0], return_counts=True))},
           "uk_ms": uk_ms, "uk_ms_headline": headline_uk_ms, "uk_ms_ratio_to_headline": uk_ms / headline_uk_ms,
           "value_ratio_to_headline": (ok * 12 * steps / elapsed) / headline_value,
           "uk_solves": int(kern[-1]["uk_solves"]), "systems_on_fp64_covariance_build": int(kern[-1]["uk_f64_solves"]),
           "fitted_variograms": {k: v for k, v in fi.items() if k != "fit_s"}, "fit_s": fi["fit_s"]}
    if not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        orc.build()
        n = 16
        cores = os.cpu_count() or 1
        r0 = c0 = max(0, args.size // 2 - n // 2)
        ref = orc.interp_grid(orc.Db(fitted), None, orc.params(), grid, nthreads=cores, rows=slice(r0, r0 + n), cols=slice(c0, c0 + n))
        got = d_norm[:, r0:r0 + n, c0:c0 + n].cpu().numpy().astype(np.float64)
        okw = ref["status"] == 0
        rec["parity_max_abs_degC"] = float(np.abs(got - ref["norm_tmin"])[:, okw].max()) if okw.any() else None
        rec["parity_cells"] = int(okw.sum())
        rec["status_equal_oracle"] = bool(np.array_equal(status[r0:r0 + n, c0:c0 + n], ref["status"]))
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)            # (the records after this one run on the headline table)
    return rec


def c4_full_record(env, args):
    """BASELINE.json configs[3] ITSELF on one GPU: the full 3250x7000 seed-7 masked grid, 12 000-station
    seed-2 tables with the 1948-2016 observations (25 203 days), every tile holding a valid cell through
    ``driver.interp_tiles_streamed`` -- kernels of tile t + 1 over the copy-out of tile t, daily int16 + normals + SE to
    pinned host memory, a discarding sink (the reference's workers hand each chunk to the writer rank, step25:177-196;
    here the sink stands where the writer would).  Shape: step25:266-314,126-185.  PCIe-inclusive by construction."""
    import datetime as dt
    from topowx_amd import _lib, driver, synth
    from topowx_amd.dates import get_days_metadata
    T = args.strip_tile
    t_s = time.perf_counter()
    if args.c4_rows and args.c4_cols:      # (tests: a cut of the same generator, as the strip records)
        grid = synth.make_grid("C3", nrows=args.c4_rows // T * T, ncols=args.c4_cols // T * T, lat_north=45.0, full_mask=False)
    else:
        grid = synth.make_grid("C3")
    days = get_days_metadata(dt.date(1948, 1, 1), dt.date(1947 + args.c4_years, 12, 31))
    nd = int(days.size)
    seed = synth.CONFIGS["C3"][5]
    tmin = synth.make_stations(grid["bbox"], args.strip_nstns, seed, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], args.strip_nstns, seed, "tmax", days, with_obs=True)
    ctx = _lib.Context(device=env.local)
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    tiles = driver.tile_list(grid["mask"], T, T)
    mine = driver.assign_tiles(tiles, 1)[0]                       # the LPT deal at world 1: its order is the run order
    if args.c4_tiles:
        mine = mine[:args.c4_tiles]
    setup_s = time.perf_counter() - t_s
    # cells for the oracle check: four per tile from four tiles -- the first and the last of the run, the tile with the FEWEST
    # valid cells (an edge of the mask) and one from the middle
    by_valid = sorted(range(len(mine)), key=lambda q: mine[q][3])
    pick_tiles = sorted({0, len(mine) - 1, by_valid[0], len(mine) // 2})
    picks = {}
    rng = np.random.default_rng(11)
    for q in pick_tiles:
        k, i, j, _ = mine[q]
        rr, cc = np.nonzero(grid["mask"][i:i + T, j:j + T])
        sel = rng.choice(rr.size, min(4, rr.size), replace=False)
        picks[k] = [(int(rr[s]), int(cc[s])) for s in sel]
    kept = {}
    acc = {"bytes": 0, "ok": 0, "cells": 0, "fixed_cells": 0, "fixed_days": 0, "fail": {}}

    def sink(k, arrays):
        acc["bytes"] += sum(v.nbytes for v in arrays.values() if hasattr(v, "nbytes"))
        st = arrays["status"]
        valid = st != -1
        acc["cells"] += int(valid.sum())
        acc["ok"] += int((st == 0).sum())
        for c, n in zip(*np.unique(st[valid & (st != 0)], return_counts=True)):
            acc["fail"][int(c)] = acc["fail"].get(int(c), 0) + int(n)
        ni = arrays["ninvalid"]
        fx = (st == 0) & (ni > 0)
        acc["fixed_cells"] += int(fx.sum())
        acc["fixed_days"] += int(ni[fx].sum())
        for (r, c) in picks.get(k, ()):
            kept[(k, r, c)] = {n: np.array(arrays[n][..., r, c]) for n in ("daily_tmin", "daily_tmax", "norm_tmin", "norm_tmax",
                                                                            "se_tmin", "se_tmax", "ninvalid", "status")}

    driver.interp_tiles_streamed(ctx, grid, mine[:1], T, T, daily=True, sink=lambda k, a: None)     # warm-up: workspace, pinned slots
    tile_ms, plog = [], {}
    t0 = time.perf_counter()
    _, secs, dev_ms = driver.interp_tiles_streamed(ctx, grid, mine, T, T, daily=True, sink=sink, tile_ms=tile_ms,
                                                   precision=args.c4_precision, log=plog)
    wall = time.perf_counter() - t0
    ms = np.array([m for _, m in tile_ms])
    units = acc["ok"] * nd * 2
    rec = {"value": units / wall, "unit": "cell-days/s (Tmin + Tmax daily int16 + normals + SE + ninvalid, streamed to pinned host memory; "
                                          "PCIe-inclusive, end to end)",
           "workload": "c4: BASELINE.json configs[3] -- the FULL %dx%d seed-7 masked grid, %d stations per variable (seed %d), %d days "
                       "(1948-01-01 .. %d-12-31), %d tiles of %dx%d in the LPT order of assign_tiles(world = 1), "
                       "driver.interp_tiles_streamed with a discarding sink" % (grid["mask"].shape + (args.strip_nstns, seed, nd, 1947 + args.c4_years, len(mine), T, T)),
           "n_gpus": 1, "wall_s": wall, "tiles": len(mine), "cells_valid": acc["cells"], "cells_ok": acc["ok"],
           "cell_days": units, "failures_by_status": {str(k): v for k, v in sorted(acc["fail"].items())},
           "cells_with_fixed_days": acc["fixed_cells"], "fixed_days": acc["fixed_days"],
           "device_only_cell_days_per_s": units / (dev_ms * 1e-3), "device_ms_total": dev_ms,
           "device_ms_per_tile": {"min": float(ms.min()), "median": float(np.median(ms)), "max": float(ms.max())},
           "d2h_bytes": acc["bytes"], "d2h_GBps": acc["bytes"] / wall / 1e9,
           "precision": {k: v for k, v in plog.items() if k != "tile_modes"},
           "stream": "the warm-up call's: driver.interp_tiles_streamed keeps its stream (device images, pinned host slots) with the context -- "
                     "pinning / unpinning 19 GB of host slots (~1 s each way; rounds 4-6 timed it with the run) is set-up, as the workspace is",
           "setup_s": setup_s}
    if not args.no_cpu_baseline and kept:
        from oracle import pyoracle as orc
        orc.build()
        dbn, dbx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
        origin = {k: (i, j) for k, i, j, _ in mine}
        worst_n, worst_lsb, flips, nvals, ninv_eq, stat_eq = 0.0, 0, 0, 0, True, True
        for (k, r, c), got in kept.items():
            i, j = origin[k]
            want = orc.interp_grid(dbn, dbx, prm, grid, daily=True, nthreads=1, rows=slice(i + r, i + r + 1), cols=slice(j + c, j + c + 1))
            stat_eq = stat_eq and int(want["status"][0, 0]) == int(got["status"])
            if want["status"][0, 0] != 0:
                continue
            ninv_eq = ninv_eq and int(want["ninvalid"][0, 0]) == int(got["ninvalid"])
            for n in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
                worst_n = max(worst_n, float(np.abs(got[n].astype(np.float64) - want[n][:, 0, 0]).max()))
            for n in ("daily_tmin", "daily_tmax"):
                dd = np.abs(got[n].astype(np.int64) - want[n][:, 0, 0].astype(np.int64))
                worst_lsb = max(worst_lsb, int(dd.max()))
                flips += int((dd != 0).sum())
                nvals += dd.size
        rec["spot_check_vs_oracle"] = {"cells": len(kept), "tiles": len(picks), "tile_valid_cells": [mine[q][3] for q in pick_tiles],
                                       "normals_max_abs_degC": worst_n, "int16_max_abs_lsb": worst_lsb,
                                       "int16_values": nvals, "int16_differing": flips, "ninvalid_equal": ninv_eq, "status_equal": stat_eq}
    ctx.drop_streams()                                            # (one stream's pinned slots at a time: 19-27 GB each)
    if not args.no_c4_deflate:
        rec["deflated_on_gpu"] = c4_deflated_record(ctx, grid, mine, T, nd, args, kept, wall, acc["bytes"], plog.get("precision"))
        ctx.drop_streams()
    if args.c4_sink_tiles > 0:
        rec["sink"] = c4_sink_record(ctx, grid, mine[:args.c4_sink_tiles], T, days, args)
    ctx.close()
    return rec


def c4_deflated_record(ctx, grid, mine, T, nd, args, kept, wall_int16, bytes_int16, int16_precision):
    """configs[3] once more, the daily values leaving the GPU as the chunk bytes of an HDF5 dataset with shuffle + deflate (the
    storage form of the reference's products, tiling.py:720,894,913,1035) formed on the device (csrc/twx_deflate.h,
    twx_stream_deflate): what crosses PCIe and what a writer has to store is ~2/3 of the int16 arrays, and no core deflates.
    Same tiles, same order, same discarding sink as the run above.  Check: the streams of the chunks holding the cells the
    run above kept for its oracle check are inflated by zlib (outside the timing) and give the same int16 series."""
    import zlib
    from topowx_amd import driver
    cy = cx = 50
    ncx = T // cx
    origin = {k: (i, j) for k, i, j, _ in mine}
    want_chunks = {}
    for (k, r, c) in kept:
        want_chunks.setdefault(k, set()).add((r // cy) * ncx + c // cx)
    blobs, acc = {}, {"bytes": 0, "stream_bytes": 0, "tiles": 0}

    def sink(k, arrays):
        acc["tiles"] += 1
        for v in ("tmin", "tmax"):
            n = sum(len(b) for b in arrays["deflated_" + v])
            acc["stream_bytes"] += n
            acc["bytes"] += n
            for ch in want_chunks.get(k, ()):
                blobs[(k, v, ch)] = bytes(arrays["deflated_" + v][ch])
        acc["bytes"] += sum(a.nbytes for a in arrays.values() if hasattr(a, "nbytes"))

    kw = dict(daily=True, deflate_chunks=(cy, cx))
    driver.interp_tiles_streamed(ctx, grid, mine[:1], T, T, sink=lambda k, a: None, **kw)       # warm-up: chunk slots, pinned stream blocks
    plog = {}
    t0 = time.perf_counter()
    _, secs, dev_ms = driver.interp_tiles_streamed(ctx, grid, mine, T, T, sink=sink, precision=args.c4_precision, log=plog, **kw)
    wall = time.perf_counter() - t0
    deflate_ms = ctx.timing()["deflate_ms"]
    int16_bytes = acc["tiles"] * 2 * nd * T * T * 2
    rec = {"wall_s": wall, "wall_s_int16_run": wall_int16, "speedup_end_to_end": wall_int16 / wall, "tiles": acc["tiles"],
           "chunks": "(%d, %d, %d) int16 -> one zlib stream each: low bytes stored, high bytes run-length coded in dynamic-Huffman blocks (one code per variable and tile)" % (nd, cy, cx),
           "d2h_bytes": acc["bytes"], "d2h_bytes_int16_run": bytes_int16, "d2h_GBps": acc["bytes"] / wall / 1e9,
           "stream_bytes_over_int16": acc["stream_bytes"] / int16_bytes, "int16_equivalent_GBps": int16_bytes / wall / 1e9,
           "device_ms_total": dev_ms, "deflate_kernels_ms_last_tile": deflate_ms,
           # the deflate kernels against their roofline (HBM): every value read by the count and the emit pass, every 16th segment by
           # the histogram pass, the streams written once (counters: profiles/r6_deflate_kernels.json)
           "deflate_kernels_roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                        "achieved": ((2 + 1 / 16.0) + acc["stream_bytes"] / int16_bytes) * 2 * nd * T * T * 2 / max(deflate_ms, 1e-9) / 1e6,
                                        "note": "algorithmic bytes of one full tile / the last tile's deflate kernel time (HIP events)"},
           "precision": {k: v for k, v in plog.items() if k != "tile_modes"}}
    # zlib -- the decoder inside libhdf5 -- on the kept cells' chunks
    ndiff, worst, n = 0, 0, 0
    for (k, r, c), got in kept.items():
        ch = (r // cy) * ncx + c // cx
        for v in ("tmin", "tmax"):
            raw = np.frombuffer(zlib.decompress(blobs[(k, v, ch)]), np.uint8)
            m = raw.size // 2
            e = np.arange(nd) * (cy * cx) + (r % cy) * cx + c % cx
            series = (raw[e].astype(np.uint16) | (raw[m + e].astype(np.uint16) << 8)).view(np.int16)
            dd = np.abs(series.astype(np.int32) - got["daily_" + v].astype(np.int32))
            ndiff += int((dd != 0).sum())
            worst = max(worst, int(dd.max()))
            n += nd
    # (the two runs may end in different precisions -- "auto" decides by each run's own device / copy-out times --: then the
    # default build's ~1e-5 of values one count off show here; the same precision gives the same bits)
    rec["deflate_kernels_roofline"]["frac"] = rec["deflate_kernels_roofline"]["achieved"] / HBM_PEAK_GBS
    rec["inflated_by_zlib"] = {"cells": len(kept), "int16_values": n, "differing_from_the_int16_run": ndiff, "max_abs_lsb": worst,
                               "precision_of_the_int16_run": int16_precision, "precision_of_this_run": plog.get("precision")}
    return rec


def host_page_rates(path, threads, gb=4.0):
    """What the host gives a writer of NEW file pages: ``threads`` workers fill disjoint parts of a fresh ``gb``-GB file under
    ``path`` through one shared mmap (cold: every page is allocated by its first touch), then once more (warm).  GB/s each --
    the ceiling of any sink into that file system, named beside the sink's own rate."""
    from concurrent.futures import ThreadPoolExecutor
    n = int(gb * 1e9) // (threads * 4096) * (threads * 4096)
    fp = os.path.join(path, "twx_page_rate.bin")
    with open(fp, "wb") as f:
        f.truncate(n)
    mm = np.memmap(fp, dtype=np.uint8, mode="r+")
    part = n // threads
    out = {}
    with ThreadPoolExecutor(threads) as pool:
        for name, val in (("cold", 1), ("warm", 2)):
            t0 = time.perf_counter()
            list(pool.map(lambda i: mm[i * part:(i + 1) * part].fill(val), range(threads)))
            out[name + "_GBps"] = n / (time.perf_counter() - t0) / 1e9
    del mm
    os.remove(fp)
    return out


def c4_sink_record(ctx, grid, tiles, T, days, args):
    """the tiles of configs[3] into the reference's per-tile NetCDF-4 files (tiling.py:304-537: chunked
    ``(ndays, 50, 50)`` int16, uncompressed as the reference's tiles are) through ``ncio.TileSink`` as the sink of
    ``driver.interp_tiles_streamed``, against the same tiles into a discarding sink.  Every tile's files are deleted once written
    (and, the first one, read back through libhdf5 and compared with the pinned block the GPU's outputs arrived in): a run of
    16 tiles is 100 GB.  Also two tiles deflated (shuffle + zlib level 1, as the reference's MOSAICS are stored).  The timed runs
    only write (files are deleted after the run when the file system has room for all of them, else by a second thread); one
    more tile per mode is written, read back through libhdf5 and compared, outside the timing."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from topowx_amd import driver, ncio
    from topowx_amd.interp import Tiler
    base = args.c4_sink_dir
    if base is None:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 20e9 else os.environ.get("TMPDIR", "/tmp")
    out_dir = os.path.join(base, "twx_c4_sink_%d" % os.getpid())
    info = Tiler(grid, T, T, 50, 50, process_tiles=()).build_tile_grid_info()
    threads = args.c4_sink_threads or min(32, os.cpu_count() or 8)
    rec = {"dir": base, "tiles": len(tiles), "threads": threads, "host_cpus": os.cpu_count(),
           "layout": "<tile_id>/<tile_id>_<var>.nc, NetCDF-4, daily int16 chunked (%d, 50, 50), normals / SE f4, inconsist_tair i4" % days.size}
    try:
        _, wall0, _ = driver.interp_tiles_streamed(ctx, grid, tiles, T, T, daily=True, sink=lambda k, a: None, precision=args.c4_precision)
        ctx.drop_streams()
        rec["wall_discarding_sink_s"] = wall0
        per_tile = 2 * days.size * T * T * 2 * 1.01
        for name, kw, sub in (("netcdf4", dict(zlib=False), tiles), ("netcdf4_deflate1", dict(zlib=True, complevel=1), tiles[:2]),
                              ("netcdf4_deflated_on_gpu", dict(zlib=True), tiles)):
            keep_all = shutil.disk_usage(base).free > 2.5 * per_tile * len(sub) + 40e9      # room for every tile of the run: delete afterwards
            writers = 1 if name == "netcdf4_deflate1" else 2     # (two tiles = four files written at once: pwrite rates add up per file)
            dkw = dict(deflate_chunks=(50, 50)) if name == "netcdf4_deflated_on_gpu" else {}     # chunk bytes formed by the GPU: the sink only appends them
            sink = ncio.TileSink(info, out_dir, days, threads=threads, order=[t[0] for t in sub], ahead=3, prep_threads=4, **kw)
            dropper = ThreadPoolExecutor(1)

            def write(k, arrays, sink=sink, keep_all=keep_all, dropper=dropper):
                sink(k, arrays)
                if not keep_all:                                 # (off the sink's thread: freeing 6 GB of pages takes about a second)
                    dropper.submit(shutil.rmtree, os.path.join(out_dir, info.get_tile_id(k)), True)
            _, wall, _ = driver.interp_tiles_streamed(ctx, grid, sub, T, T, daily=True, sink=write, precision=args.c4_precision,
                                                      writer_threads=writers, **dkw)
            sink.close()
            dropper.shutdown(wait=True)
            ctx.drop_streams()
            st = dict(sink.stats)
            shutil.rmtree(out_dir, ignore_errors=True)
            # one more tile, outside the timing: written, read back through libhdf5, compared with the pinned block it came from
            chk = ncio.TileSink(info, out_dir, days, threads=threads, verify=(sub[0][0],), **kw)
            driver.interp_tiles_streamed(ctx, grid, sub[:1], T, T, daily=True, sink=chk, precision=args.c4_precision, **dkw)
            chk.close()
            ctx.drop_streams()
            shutil.rmtree(out_dir, ignore_errors=True)
            rec[name] = {"tiles": st["tiles"], "wall_s": wall, "int16_GB": st["int16_bytes"] / 1e9, "on_disk_GB": st["disk_bytes"] / 1e9,
                         "int16_GBps_end_to_end": st["int16_bytes"] / wall / 1e9, "on_disk_GBps_end_to_end": st["disk_bytes"] / wall / 1e9,
                         "sink_calls_in_flight": writers,
                         "sink_busy_s": st["total_s"], "of_it_waiting_for_prepared_files_s": st["prepare_s"], "of_it_bulk_copy_s": st["copy_s"],
                         "posix_fallocate_s_off_thread": st["fallocate_s"], "look_ahead": st["look_ahead"],
                         "int16_GBps_while_sink_busy": st["int16_bytes"] / max(st["total_s"], 1e-9) / 1e9,
                         "files_kept_until_the_end_of_the_run": keep_all, "tiles_read_back_equal": chk.stats["verified"]}
        rec["host_new_page_rate"] = host_page_rates(base, threads)
        n4 = rec["netcdf4"]
        rec["limiting_stage"] = ("the sink: %.1f GB/s of int16 into NetCDF-4 tile files (pipeline start-up included) against %.1f GB/s into a "
                                 "discarding sink.  New file pages by first touch come at %.1f GB/s on this host (one file, %d threads: they "
                                 "serialise on the file's page-cache lock), so the sink allocates a file's pages with ONE posix_fallocate "
                                 "(%.1f GB/s per file here) on look-ahead threads, gathers a tile into a warm staging buffer in chunk order "
                                 "and pwrites the chunks -- ~11 GB/s per file, files add up -- with %d tiles in flight"
                                 % (n4["int16_GBps_end_to_end"], n4["int16_GB"] / max(wall0, 1e-9),
                                    rec["host_new_page_rate"]["cold_GBps"], threads,
                                    n4["on_disk_GB"] / max(n4["posix_fallocate_s_off_thread"], 1e-9), n4["sink_calls_in_flight"]))
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    return rec


def config5_record(env, args):
    """BASELINE.json configs[4]: leave-one-out cross-validation + bandwidth optimisation over all stations."""
    from topowx_amd import xval
    res, arr = xval.run_config5(args.c5_nstns, args.c5_years, "tmin", 0, 0, 1, env.local, "cpu", db="c5")
    rec = dict(res)
    rec["workload"] = ("c5: step21 (variogram fit + kriging x 16 bandwidths), step22 (every station's variogram with the optimised "
                       "bandwidths), step23 (GWR series + statistics x 16 bandwidths), step24 (normals + daily) over all %d "
                       "cross-validated stations of the %d-station seed-2 database drawn over the configs[2] grid (SURVEY 8d), "
                       "12 months, %d years of days; bandwidth optimisation (optimize.py:268-374) after step21 / step23"
                       % (res["stations"], res["stations_in_db"], args.c5_years))
    if not args.no_cpu_baseline:
        # step21's leave-one-out errors of three stations x three bandwidths against the oracle
        from oracle import pyoracle as orc
        from topowx_amd import stationdb as sdb
        orc.build()
        stn = arr["stn"]
        before = sdb.StationSerialDataDb(arr["stns_step21"], "tmin", stn.days, None)
        db, prm = orc.Db(before), orc.params()
        c = db.cols
        good = np.isnan(before.stns[sdb.BAD])
        idx = {s: i for i, s in enumerate(before.stns[sdb.STN_ID][good])}
        worst, n = 0.0, 0
        for q in np.random.default_rng(5).choice(arr["ids"].size, 3, replace=False):
            j = idx[arr["ids"][q]]
            pt = orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
            for x in (0, 7, 13):
                rc, want, _ = orc.krigall(db, prm, pt, int(xval.DFLT_LADDER[x]), excl=j, rm_zero_dist=True)
                if rc == 0 and np.isfinite(arr["mae_norm"][:, x, q]).all():
                    worst = max(worst, float(np.abs(arr["mae_norm"][:, x, q] - np.abs(want - c["norm"][:, j])).max()))
                    n += 12
        rec["spot_check_vs_oracle"] = {"step21_values": n, "max_abs_degC": worst}
    return rec


# =====================================================================================================================
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(args)

    env = Env()
    torch = env.torch
    from topowx_amd import _lib, synth
    world, rank, dev = env.world, env.rank, env.dev
    scaling = args.scaling
    top_strong = scaling == "strong"

    if top_strong:
        rec = strip_run(env, args, args.steps, args.warmup, spot_check=world == 1)
        if rank == 0:
            res = {"metric": "grid-cell-days interpolated/sec", "value": rec["value"],
                   "unit": "cell-months/s (normals config: one time step = one calendar month, mean + SE)",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                   "config": {"workload": rec["workload"], "cells_ok": rec["cells_ok"],
                              "parallelism": "tiles of ONE grid dealt over %d GPU(s) (LPT), station table replicated, "
                                             "normals mosaic gathered on rank 0" % world},
                   "roofline": {"bound": "hbm", "achieved": rec["hbm"]["achieved_GBps"], "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                "frac": rec["hbm"]["frac_of_peak"], "traffic": None,
                                "note": "whole-job figure (all ranks); the per-kernel roofline is in the N = 1 headline line"},
                   "strong": rec}
            if env.shared:
                res["note"] = "control-flow run: %d ranks share ONE GPU over gloo (fewer GPUs than ranks); not a scaling figure" % world
            print(json.dumps(res), flush=True)
        if world > 1:
            env.dist.destroy_process_group()
        return

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

    ctx = _lib.Context(device=env.local)
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

    kern = []
    elapsed = timed(env, lambda: ctx.interp_grid_dev(g, o, _lib.VAR_TMIN_BIT, stream), args.steps, args.warmup,
                    lambda keep: kern.append(ctx.timing()) if keep else ctx.timing())   # HIP events on the launch stream (synchronises the step)
    status = d_stat.cpu().numpy()
    ncell_ok = int((status == 0).sum())
    units_per_step = ncell_ok * 12                     # (cell, month) outputs, each mean + SE
    value = world * units_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernels (universal kriging) ---------------------------------
    uk_ms = float(np.mean([t["uk_ms"] for t in kern]))
    launches = max(1, int(kern[-1]["uk_launches"]))
    solves = int(kern[-1]["uk_solves"])
    f64_solves = int(kern[-1]["uk_f64_solves"])
    ach_gbs = ALG_BYTES_PER_CELL_MONTH * solves / (uk_ms * 1e-3) / 1e9
    # bandwidths actually used by the timed steps (diagnostic accessor, no extra launches)
    ks = ctx.last_bandwidths(_lib.TMIN).ravel()
    kpos = ks[ks > 0]
    flops_per_solve = float(uk_flops(kpos).mean())
    flops_exec = float(uk_flops_executed(kpos).mean())
    # systems per kriging launch (twx_krig_bucket, twx_select.h): largest k of the bucket -> count
    edges = np.array([40, 48, 56, 64, 72, 80, 88, 96, 104, 120, 136, 152])
    bidx = np.searchsorted(edges, kpos)
    rows_hist = np.bincount(bidx, minlength=edges.size + 1)
    ach_tflops = flops_per_solve * solves / (uk_ms * 1e-3) / 1e12
    ach_tflops_exec = flops_exec * solves / (uk_ms * 1e-3) / 1e12
    # PMC counters cannot be read from inside the process: the committed measurement of THIS workload is quoted
    traffic, daily_traffic, traffic_src = (None, None, None)
    if world == 1 and args.size == 250 and args.nstns == 10000:
        traffic, daily_traffic, traffic_src = latest_traffic()

    res = {
        "metric": "grid-cell-days interpolated/sec",
        "value": value,
        "unit": "cell-months/s (normals config: one time step = one calendar month, mean + SE)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE, "dtype_note": DTYPE_NOTE, "data": "synthetic",
        "config": {"workload": "C2: one %dx%d 30-arcsec tile per GPU, %d synthetic stations, 12 monthly Tmin "
                               "normals + SE (BASELINE.json configs[1])" % (Y, X, ctx.nstn[_lib.TMIN]),
                   "cells_ok": ncell_ok, "mean_nnghs": float(kpos.mean()),
                   "parallelism": "tiles partitioned over %d GPU(s), station table replicated" % world},
        "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                     "traffic_measured_in_this_run": False,   # PMC counters cannot be read from inside the process
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": ALG_BYTES_PER_CELL_MONTH * solves / launches,
                     "kernel": "universal kriging: k_tile_dist + k_ukw<..> + k_uk<..> (%d launches per step)" % launches,
                     "kernel_ms_per_step": uk_ms, "systems_on_fp64_covariance_build": f64_solves,
                     "note": "path is fp64-VALU bound, not HBM bound (SURVEY.md 8d); see fp64"},
        "fp64": {"achieved": ach_tflops, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach_tflops / FP64_VEC_PEAK_TFLOPS, "flops_per_solve": flops_per_solve,
                 # what the fraction is made of: the nominal figure counts ~60 flops per station pair of EVERY system
                 # (SURVEY 8d); the kernels evaluate each pair once per tile (k_tile_dist), so that term is not executed
                 "frac_executed": ach_tflops_exec / FP64_VEC_PEAK_TFLOPS, "achieved_executed": ach_tflops_exec,
                 "flops_per_solve_executed": flops_exec,
                 "distance_flops_share": float(uk_flops_distance(kpos).mean()) / flops_per_solve,
                 "note": "frac = SURVEY 8d's nominal flops (k^3/3 + 7k^2 + 30k(k-1)) / kernel time / peak; frac_executed counts "
                         "k^3/3 + 7k^2 only (Cholesky + border rows)",
                 "systems_by_bucket_kmax": {str(int(e)): int(c) for e, c in zip(edges, rows_hist) if c},
                 # executed flops (k^3/3 + 7k^2) of a step's systems per bucket: tests/tools/reduce_sq.py divides the fp64
                 # FMAs a kernel ISSUED (SQ counters) by the FMAs these flops NEED (128 flops per wave instruction)
                 "executed_flops_by_bucket_kmax": {str(int(e)): float(uk_flops_executed(kpos[bidx == i]).sum())
                                                   for i, e in enumerate(edges) if (bidx == i).any()},
                 "kernels_by_bucket_kmax": {"40": "k_ukw2<3,0>", "48": "k_ukwz<3,0>", "56": "k_ukw<4,0>", "64": "k_ukwz<4,0>", "72": "k_ukw<5,0>",
                                            "80": "k_ukwz<5,0>", "88": "k_ukw<6,0>", "96": "k_ukwz<6,0>", "104": "k_uk<7,2,0>",
                                            "120": "k_uk<8,4,0>", "136": "k_uk<9,2,0>", "152": "k_uk<10,2,0>"}},
        "timing_ms": {k: float(np.mean([t[k] for t in kern])) for k in
                      ("tile_cand_ms", "select_ms", "uk_ms", "total_ms")},
    }
    if env.shared:
        res["note"] = "control-flow run: %d ranks share ONE GPU over gloo (fewer GPUs than ranks); not a scaling figure" % world

    import datetime as dt
    # ---- daily record: the path that produces cell-DAYS (N = 1) --------------------------------------
    if world == 1 and not args.no_daily:
        res["daily"] = daily_record(env, args, base, grid, g, d_ninv, d_stat, dt.date(args.daily_year0, 1, 1),
                                    dt.date(args.daily_year0 + args.daily_years - 1, 12, 31),
                                    "C2 tile, %d years from %d" % (args.daily_years, args.daily_year0),
                                    args.steps, args.warmup, args.stream_tiles, args.int16_window,
                                    daily_traffic if args.daily_years == 10 else None, traffic_src)

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

    # ---- the other BASELINE.json configurations, driver-timed in the same line (N = 1) ----------------
    if world == 1 and not args.no_configs and (args.size == 250 or args.force_configs):
        want_cfg = [c for c in args.configs.split(",") if c]
        cfg = {}
        if "c2_fitted" in want_cfg:
            t1 = time.perf_counter()
            cfg["c2_fitted"] = c2_fitted_record(env, args, ctx, stn, grid, g, o, d_norm, d_stat, stream, uk_ms, value)
            cfg["c2_fitted"]["record_wall_s"] = time.perf_counter() - t1
        if "c4_tile" in want_cfg:
            t1 = time.perf_counter()
            cfg["c4_tile"] = daily_record(env, args, base, grid, g, d_ninv, d_stat, dt.date(1948, 1, 1), dt.date(2016, 12, 31),
                                          "c4_tile (BASELINE.json configs[3], one tile)", 3, 1, 0, 2)
            cfg["c4_tile"]["record_wall_s"] = time.perf_counter() - t1
            # measured HBM bytes of the daily launches on THIS tile (tests/tools/collect_c4_traffic.sh): the hat rows are per
            # (cell, month), so their share shrinks with the day axis -- quoted from the newest committed PMC reduction
            c4p = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c4_daily_traffic.json")))
            if c4p and args.size == 250 and args.nstns == 10000:
                tj = json.load(open(c4p[-1]))
                cfg["c4_tile"]["traffic"] = {"bytes_per_step": tj["fetched_bytes_per_step"] + tj["written_bytes_per_step"],
                                             "algorithmic_bytes_per_step": tj["algorithmic_bytes_per_step"], "ratio": tj["ratio"],
                                             "measured_in_this_run": False, "source": "profiles/" + os.path.basename(c4p[-1])}
        ctx.close()
        ctx = None
        del d_in, d_norm, d_se
        torch.cuda.empty_cache()
        if "c5" in want_cfg:
            t1 = time.perf_counter()
            cfg["c5"] = config5_record(env, args)
            cfg["c5"]["record_wall_s"] = time.perf_counter() - t1
        if "c3" in want_cfg:
            t1 = time.perf_counter()
            cfg["c3"] = strip_run(env, args, args.c3_steps, 1, spot_check=True, full=True)
            cfg["c3"]["record_wall_s"] = time.perf_counter() - t1
        if "c3_fitted" in want_cfg:
            t1 = time.perf_counter()
            cfg["c3_fitted"] = strip_run(env, args, 1, 1, spot_check=True, full=not args.force_configs, fitted=True)   # (tests: the strip)
            cfg["c3_fitted"]["record_wall_s"] = time.perf_counter() - t1
        if "c3_strip" in want_cfg:
            t1 = time.perf_counter()
            cfg["c3_strip"] = strip_run(env, args, args.strong_steps, 1, spot_check=True)
            cfg["c3_strip"]["record_wall_s"] = time.perf_counter() - t1
        if "c4" in want_cfg:
            t1 = time.perf_counter()
            torch.cuda.empty_cache()
            cfg["c4"] = c4_full_record(env, args)
            cfg["c4"]["record_wall_s"] = time.perf_counter() - t1
        res["configs"] = cfg
    if ctx is not None:
        ctx.close()
        ctx = None

    # ---- N > 1: the tile farm on one fixed grid next to the weak headline ------------------------------
    if world > 1 and scaling == "auto":
        del d_in, d_norm, d_se
        torch.cuda.empty_cache()
        rec = strip_run(env, args, args.strong_steps, 1, spot_check=False)
        rec["scaling"] = "strong"
        rec["note"] = ("total work fixed as N grows; its N = 1 counterpart: python bench.py --configs c3_strip (same code "
                       "path as configs.c3 of the N = 1 line: topowx_amd.driver)")
        if args.strong_daily_rows > 0:
            rec["daily"] = strong_daily_run(env, args)
        res["strong"] = rec
    if rank == 0:
        print(json.dumps(res), flush=True)
    if env.pg:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
