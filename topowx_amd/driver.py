"""Python-3 counterpart of ``scripts/step25_mpi_interp_tair.py`` for one node of GPUs.

The reference farms 50x50 work chunks of 250x250 tiles over MPI workers that each
loop over cells in Python and write their own netCDF chunk (step25:49-314).  Here
one process drives one GPU (``torch.distributed``, backend nccl = RCCL on ROCm,
gloo in the CPU tests): tiles are dealt to ranks balanced by their number of
unmasked cells, every rank holds the full station table (a few MB; the obs matrix
~1 GB per variable) and interpolates whole tiles with one library call each.
Cells are independent, so the data path needs NO collective; the optional
``gather_mosaic_device`` assembles the f4 normals / SE mosaics on rank 0 over xGMI
straight from the device tensor the tiles were computed into (``interp_tiles_device``;
``gather_mosaic`` is the host-array form the streamed / daily paths use) (SURVEY.md
section 8e).  Results are returned as arrays / written as ``.npz``; the netCDF tile
writer is ``topowx_amd.ncio`` / ``topowx_amd.step25``.
"""
import argparse
import json
import os
import time

import numpy as np

FILL_F4 = np.float32(9.969209968386869e36)


def tile_list(mask, tile_y, tile_x):
    """[(tile number, row0, col0, n valid cells)] for tiles holding >= 1 valid cell,
    numbered like the reference (row-major over tiles, empty tiles skipped: tiling.py:131-165)."""
    mask = np.asarray(mask) != 0
    out, k = [], 0
    for i in range(0, mask.shape[0], tile_y):
        for j in range(0, mask.shape[1], tile_x):
            n = int(mask[i:i + tile_y, j:j + tile_x].sum())
            if n > 0:
                out.append((k, i, j, n))
                k += 1
    return out


def assign_tiles(tiles, world):
    """Deterministic longest-processing-time deal: heaviest tile to the least loaded rank."""
    load = [0] * world
    mine = [[] for _ in range(world)]
    for t in sorted(tiles, key=lambda t: (-t[3], t[0])):
        r = min(range(world), key=lambda q: (load[q], q))
        mine[r].append(t)
        load[r] += t[3]
    for m in mine:
        m.sort(key=lambda t: t[0])
    return mine


def interp_tiles(grid, compute, tiles, tile_y, tile_x):
    """Run ``compute(grid, rows, cols) -> dict of arrays`` on every tile of this rank."""
    return {k: compute(grid, slice(i, i + tile_y), slice(j, j + tile_x)) for k, i, j, _ in tiles}


def gather_mosaic(local, assignment, shape, tile_y, tile_x, keys, rank, world, device="cpu"):
    """Assemble [12, Y, X] f4 mosaics of ``keys`` on rank 0.

    One equal-size (padded) tensor per rank through ``dist.gather``; over RCCL every
    peer has a direct xGMI link to the root, so this is world-1 concurrent
    point-to-point transfers.
    """
    import torch
    import torch.distributed as dist
    nmax = max(len(a) for a in assignment)
    buf = np.full((nmax, len(keys), 12, tile_y, tile_x), FILL_F4, np.float32)
    for s, (k, _, _, _) in enumerate(assignment[rank]):
        for q, key in enumerate(keys):
            a = local[k][key]                                  # an edge tile may be smaller than tile_y x tile_x
            buf[s, q, :, :a.shape[1], :a.shape[2]] = a
    t = torch.from_numpy(buf).to(device)
    if world > 1:
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, parts, dst=0)
    else:
        parts = [t]
    if rank != 0:
        return None
    mosaic = {key: np.full((12,) + tuple(shape), FILL_F4, np.float32) for key in keys}
    for r in range(world):
        arr = parts[r].cpu().numpy()
        for s, (k, i, j, _) in enumerate(assignment[r]):
            for q, key in enumerate(keys):
                m = mosaic[key][:, i:i + tile_y, j:j + tile_x]    # clipped at the grid edge
                m[...] = arr[s, q][:, :m.shape[1], :m.shape[2]]
    return mosaic


NORMAL_KEYS = ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax")


def upload_grid(grid, device, tiles=None, tile_y=None, tile_x=None):
    """The predictor planes as device tensors (native dtypes, tiling.py:190-213).  With ``tiles`` (a rank's share of
    ``assign_tiles``) only THOSE tiles' planes go up, each as its own contiguous image: 1 / world of the grid per rank (the
    full configs[2] grid is 2.48 GB; a rank's 40 of 323 tiles 0.31 GB) instead of a replica on every GPU."""
    import torch
    from . import _lib
    if tiles is None:
        a = _lib.Context.grid_arrays(grid)
        return {k: torch.from_numpy(v).to(device) for k, v in a.items()}
    Yg, Xg = np.asarray(grid["mask"]).shape
    cut = {}
    for (k, i, j, _) in tiles:
        a = _lib.Context.grid_arrays(grid, slice(i, min(i + tile_y, Yg)), slice(j, min(j + tile_x, Xg)))
        cut[k] = {n: torch.from_numpy(np.ascontiguousarray(v)).to(device) for n, v in a.items()}
    return {"tiles": cut, "shape": (Yg, Xg), "device": torch.device(device)}


def interp_tiles_device(ctx, dgrid, tiles, tile_y, tile_x, variables=("tmin", "tmax"), nslots=None, stream=None, stats=None):
    """Normals + SE of this rank's tiles with everything resident in HBM: the predictor planes come from the device
    tensors of ``upload_grid`` (whole grid: each tile's 61 B / cell are gathered into a contiguous image by a device copy;
    per-tile upload: used as they are), and ``twx_interp_grid_dev`` writes every tile's outputs straight into slot s of ONE
    device tensor ``buf[nslots, 4, 12, tile_y, tile_x]`` -- the send buffer of ``gather_mosaic_device``; nothing crosses
    PCIe.  ``nslots`` >= len(tiles) pads the buffer to the size every rank of a gather must share.  The loop does not
    wait for a tile before it enqueues the next one (the library's own 64-byte read-back per call apart): per-tile times
    come from events recorded on the launch stream and are read once, after the loop.  ``stats``: a dict that accumulates the
    library's launch statistics (``uk_solves``, ``uk_f64_solves``) over the tiles -- a diagnostic: it reads them after every
    tile, which waits for the tile.
    Returns (buf, status[nslots, tile_y, tile_x] i4 device tensor, per-tile device ms)."""
    import torch
    from . import _lib
    per_tile = "tiles" in dgrid
    dev = dgrid["device"] if per_tile else dgrid["mask"].device
    nslots = len(tiles) if nslots is None else nslots
    assert nslots >= len(tiles)
    buf = torch.full((max(nslots, 1), 4, 12, tile_y, tile_x), float(FILL_F4), dtype=torch.float32, device=dev)
    stat = torch.full((max(nslots, 1), tile_y, tile_x), -1, dtype=torch.int32, device=dev)
    ninv = torch.empty((tile_y, tile_x), dtype=torch.int32, device=dev)
    Yg, Xg = dgrid["shape"] if per_tile else dgrid["mask"].shape
    vars_mask = (_lib.VAR_TMIN_BIT if "tmin" in variables else 0) | (_lib.VAR_TMAX_BIT if "tmax" in variables else 0)
    tstream = torch.cuda.current_stream() if stream is None else torch.cuda.ExternalStream(stream)
    strm = tstream.cuda_stream
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(tiles) + 1)]
    marks[0].record(tstream)
    # everything of a tile -- the gathers of its input planes, the edge tile's scratch images, the library call, the placement of
    # an edge tile -- runs on ONE stream (the caller's `stream` when given): torch's copies are ordered with the kernels that read /
    # write the same tensors, and the caching allocator recycles t / o / so only behind work queued on that stream
    tstream.wait_stream(torch.cuda.current_stream())        # (buf / stat / the uploaded planes were filled on the current stream)
    with torch.cuda.stream(tstream):
        for s, (k, i, j, _) in enumerate(tiles):
            y, x = min(tile_y, Yg - i), min(tile_x, Xg - j)
            if per_tile:
                t = dgrid["tiles"][k]
            else:
                t = {n: dgrid[n][i:i + y, j:j + x].contiguous() for n in ("mask", "elev", "tdi", "climdiv")}
                t["lat"] = dgrid["lat"][i:i + y].contiguous()
                t["lon"] = dgrid["lon"][j:j + x].contiguous()
                for n in ("lst_night", "lst_day"):
                    t[n] = dgrid[n][:, i:i + y, j:j + x].contiguous()
            full = y == tile_y and x == tile_x
            # an edge tile is smaller than its slot: computed into a scratch image and placed afterwards
            o = buf[s] if full else torch.full((4, 12, y, x), float(FILL_F4), dtype=torch.float32, device=dev)
            so = stat[s] if full else torch.full((y, x), -1, dtype=torch.int32, device=dev)
            g = _lib.TwxGrid(y, x, *[t[n].data_ptr() for n in ("mask", "lat", "lon", "elev", "tdi", "climdiv", "lst_night", "lst_day")])
            ptr = [o[q].data_ptr() if v in variables else None for q, v in enumerate(("tmin", "tmin", "tmax", "tmax"))]
            go = _lib.TwxGridOut(ptr[0], ptr[1], ptr[2], ptr[3], None, None, ninv.data_ptr(), so.data_ptr())
            ctx.interp_grid_dev(g, go, vars_mask, strm)     # (stream-ordered: the tile's input images may be freed by torch afterwards)
            if stats is not None:
                t_ = ctx.timing()
                for key in ("uk_solves", "uk_f64_solves"):
                    stats[key] = stats.get(key, 0) + int(t_[key])
            if not full:
                buf[s, :, :, :y, :x] = o
                stat[s, :y, :x] = so
            marks[s + 1].record(tstream)
    marks[-1].synchronize()
    ms = [marks[s].elapsed_time(marks[s + 1]) for s in range(len(tiles))]
    return buf, stat, ms


def gather_mosaic_device(buf, assignment, shape, tile_y, tile_x, rank, world, keys=NORMAL_KEYS, backend="nccl", collective=None):
    """``gather_mosaic`` on device tensors: ``buf[nmax, len(keys), 12, tile_y, tile_x]`` of every rank (the tensor
    ``interp_tiles_device`` has filled, same nmax everywhere) goes to rank 0 with ONE ``dist.gather`` -- over RCCL
    world - 1 concurrent xGMI transfers into rank 0's HBM -- and is placed into ``[12, Y, X]`` mosaics by device
    copies; no host staging (with the gloo backend of the CPU / shared-GPU control-flow runs the collective itself
    travels through host memory).  ``collective``: run the ``dist.gather`` (default: whenever world > 1; True at world 1 sends
    rank 0's buffer through a one-rank RCCL communicator -- tests/test_gpu_rccl_smoke.py).
    Returns {key: device tensor} on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if (world > 1) if collective is None else collective:
        send = buf if backend == "nccl" else buf.cpu()
        parts = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, parts, dst=0)
    else:
        parts = [buf]
    if rank != 0:
        return None
    dev = buf.device
    Y, X = shape
    mosaic = {key: torch.full((12, Y, X), float(FILL_F4), dtype=torch.float32, device=dev) for key in keys}
    for r in range(world):
        arr = parts[r].to(dev)
        for s, (_, i, j, _) in enumerate(assignment[r]):
            y, x = min(tile_y, Y - i), min(tile_x, X - j)
            for q, key in enumerate(keys):
                mosaic[key][:, i:i + y, j:j + x] = arr[s, q, :, :y, :x]
    return mosaic


class PrecisionPolicy(object):
    """Which covariance build a streamed run uses (``Context.set_precision``), decided once per run.

      "fast"   the default routing: fp32 pair distances, only ill-conditioned systems and tie-guard cells on the fp64 build
               (normals within ~2e-6 degC of an fp64 evaluation, ~2e-5 of the packed int16 days one count off);
      "exact"  every system on the fp64 build: outputs equal an fp64 evaluation of the reference's formulas to the last
               int16 / f4 bit (tests/tools/gpu_full_tile_parity.py --f64), ~1.3 x the kriging time;
      "auto"   "exact" for as long as it is FREE: a streamed run is bound by the copy-out of its outputs whenever they are
               large (daily tiles: the GPU idles ~70 % of the wall), and then the fp64 build costs no wall time.  The run
               starts exact; the device times of the first PROBE exact tiles are compared with their copy-out times
               (``TileStream.times``), and if their MEDIAN is not hidden behind the copy (device > 0.9 x copy: normals-only
               tiles) the rest of the run is fast.  (The median, not any one tile: the first tile of a new size grows the
               library's workspace -- a hipFree + hipMalloc of gigabytes shows as half a second of "device time" once.)

    ``mode`` is the build the NEXT submitted tile gets; ``observe`` is fed every finished tile; ``summary()`` is what the tile
    logs record.  ``close()`` leaves the context in the "fast" mode."""
    PROBE = 3                                   # exact tiles looked at before "auto" decides

    def __init__(self, ctx, requested="auto"):
        if requested not in ("auto", "fast", "exact"):
            raise ValueError("precision must be 'auto', 'fast' or 'exact'")
        self.ctx, self.requested = ctx, requested
        self.mode = "fast" if requested == "fast" else "exact"
        self.decided = requested != "auto"
        self.decision = "as requested" if self.decided else "exact throughout: every probed tile's kernels were hidden behind its copy-out"
        self.seen = {"exact": [0, 0.0, 0.0], "fast": [0, 0.0, 0.0]}     # tiles, device ms, copy ms by the mode they ran in
        self.tile_modes = {}
        self._probe = []                        # (device ms, copy ms) of the exact tiles seen so far
        ctx.set_precision(self.mode)

    def observe(self, tile_mode, device_ms, copy_ms, tile=None):
        if tile is not None:
            self.tile_modes[tile] = tile_mode
        rec = self.seen[tile_mode]
        rec[0] += 1; rec[1] += device_ms; rec[2] += copy_ms
        if self.decided or tile_mode != "exact":
            return
        self._probe.append((device_ms, copy_ms))
        if len(self._probe) < self.PROBE:
            return
        self.decided = True
        dev, cp = sorted(p[0] for p in self._probe)[self.PROBE // 2], sorted(p[1] for p in self._probe)[self.PROBE // 2]
        if dev > 0.9 * cp:
            self.mode = "fast"
            self.decision = ("fast after %d tiles: the kernels of an exact tile (median %.2f ms) are not hidden behind its copy-out "
                             "(median %.2f ms)" % (self.seen["exact"][0] + self.seen["fast"][0], dev, cp))
            self.ctx.set_precision("fast")
        else:
            self.decision = ("exact throughout: the kernels of an exact tile (median %.2f ms) hide behind its copy-out (median %.2f ms)"
                             % (dev, cp))

    def summary(self):
        nt = self.seen["exact"][0] + self.seen["fast"][0]
        return {"requested": self.requested, "precision": self.mode, "tiles_exact": self.seen["exact"][0],
                "tiles_fast": self.seen["fast"][0], "device_ms_mean": (self.seen["exact"][1] + self.seen["fast"][1]) / max(nt, 1),
                "copy_ms_mean": (self.seen["exact"][2] + self.seen["fast"][2]) / max(nt, 1), "decision": self.decision,
                "tile_modes": dict(self.tile_modes)}

    def close(self):
        self.ctx.set_precision("fast")


def interp_tiles_streamed(ctx, grid, tiles, tile_y, tile_x, variables=("tmin", "tmax"), daily=False, sink=None,
                          writer_threads=1, tile_ms=None, precision="auto", log=None, deflate_chunks=None, trace=None):
    """Tiles of this rank through a ``TileStream`` (twx_stream_*): while the GPU interpolates tile t + 1 the outputs of
    tile t arrive in pinned host memory and go to ``sink(tile_number, arrays)`` on a writer thread (the reference's
    workers hand every finished chunk to a writer, step25:177-196).  ``sink`` must be done with the arrays when it
    returns (they are views of a pinned slot that is reused two tiles later); default: collect copies.  ``writer_threads`` > 1:
    that many sink calls may run at once (on as many pinned slots more) -- for a sink that is thread-safe and whose rate grows with
    the number of tiles in flight (``ncio.TileSink``: new file pages come per FILE); tiles then reach the sink in order but may
    finish out of order.  All tiles must have the shape tile_y x tile_x.  ``tile_ms``: a list that receives ``(tile_number, device_ms)`` per tile.

    ``precision``: "auto" | "fast" | "exact", see ``PrecisionPolicy`` -- auto = the fp64 covariance build (outputs exact to the
    last int16 / f4 bit) whenever the tiles' kernels hide behind their copy-out, i.e. for free.  ``log``: a dict that receives
    the policy's summary (the mode the run ended in, tiles per mode, mean device / copy ms, the decision in words).
    ``deflate_chunks=(cy, cx)``: the daily values leave the GPU as the chunk bytes of an HDF5 dataset with shuffle + deflate
    (``TileStream``, twx_stream_deflate): ``arrays`` then holds ``deflated_tmin`` / ``deflated_tmax`` (one zlib stream per
    ``(ndays, cy, cx)`` chunk, row-major chunk order) instead of the daily arrays -- about half the bytes over PCIe and no
    deflate on the host (``ncio.TileSink(zlib=True)`` appends them with ``H5Dwrite_chunk``).
    ``trace``: a list that receives ``(tile_number, "submit" | "wait", t_begin, t_end)`` in seconds since the start of the run
    (a diagnostic: where the host spends a tile's period).
    The stream (two device images, ``2 + writer_threads`` pinned host slots: 6.3 GB each for a configs[3] tile) stays with the
    context for the next run over tiles of this shape; ``ctx.drop_streams()`` releases it.
    Returns (results or None, seconds, device_ms)."""
    import queue
    import threading
    import time
    policy = PrecisionPolicy(ctx, precision)
    collected = {}
    if sink is None:
        def sink(k, arrays):
            collected[k] = {n: (np.array(v) if hasattr(v, "shape") else [bytes(b) for b in v])
                            for n, v in arrays.items() if hasattr(v, "shape") or n.startswith("deflated_")}
    writer_threads = max(1, int(writer_threads))
    nslots = 2 + writer_threads                 # one computing, one copying out, one at each writer
    # (kept with the context: the next run over tiles of this shape reuses the device images and the pinned slots)
    st = ctx.stream(tile_y, tile_x, variables=variables, daily=daily, nslots=nslots, deflate_chunks=deflate_chunks, keep=True)
    q = queue.Queue(maxsize=1)
    free = [threading.Semaphore(1) for _ in range(nslots)]
    err = []

    def writer():
        while True:
            item = q.get()
            if item is None:
                return
            k, slot, arrays = item
            try:
                sink(k, arrays)
            except Exception as e:              # noqa: BLE001 -- reported to the caller below
                err.append(e)
            finally:
                free[slot].release()

    ths = [threading.Thread(target=writer, daemon=True) for _ in range(writer_threads)]
    for th in ths:
        th.start()
    t0 = time.perf_counter()
    dev_ms = 0.0
    pending = None

    seen_in, seen_out, done = [], [], False

    def collect(pk, pslot, pmode):
        nonlocal dev_ms
        tb = time.perf_counter()
        out = st.wait(pslot)                    # tile t is on the host; tile t + 1 is already running
        seen_out.append(pk)
        if trace is not None:
            trace.append((pk, "wait", tb - t0, time.perf_counter() - t0))
        ms = out.pop("device_ms")
        dev_ms += ms
        policy.observe(pmode, ms, st.times(pslot)[1], tile=pk)
        if tile_ms is not None:
            tile_ms.append((pk, ms))
        q.put((pk, pslot, out))

    try:
        for n, (k, i, j, _) in enumerate(tiles):
            slot = n % nslots
            free[slot].acquire()                # the writer is done with this slot's previous tile
            if err:                             # the sink failed: no point in interpolating the rest
                break
            submitted_as = policy.mode
            tb = time.perf_counter()
            st.submit(slot, grid, slice(i, i + tile_y), slice(j, j + tile_x))
            seen_in.append(k)
            if trace is not None:
                trace.append((k, "submit", tb - t0, time.perf_counter() - t0))
            if pending is not None:
                collect(*pending)
            pending = (k, slot, submitted_as)
        if pending is not None:
            collect(*pending)
        done = True
    finally:
        for _ in ths:
            q.put(None)
        for th in ths:
            th.join()
        if not done or len(seen_out) != len(seen_in):     # an error on the way, a tile submitted and never collected: start afresh next time
            st.close()
        policy.close()
    if log is not None:
        log.update(policy.summary())
    if err:
        raise err[0]
    return (collected if collected else None), time.perf_counter() - t0, dev_ms


def gpu_compute(ctx, variables=("tmin", "tmax"), daily=False):
    """compute() backed by libtwxhip (the product path)."""
    def compute(grid, rows, cols):
        return ctx.interp_grid(grid, variables=variables, daily=daily, rows=rows, cols=cols)
    return compute


def main():
    ap = argparse.ArgumentParser(description="interpolate a synthetic grid on the GPUs of one node")
    ap.add_argument("--config", default="C1")
    ap.add_argument("--tile", type=int, default=50)
    ap.add_argument("--daily", action="store_true")
    ap.add_argument("--gather", action="store_true", help="assemble the normals mosaic on rank 0 (RCCL gather)")
    ap.add_argument("--out", default=None, help=".npz for rank 0's mosaic")
    ap.add_argument("--tile-dir", default=None, help="write every tile of this rank as <dir>/tileNNNNN.npz while the next one runs")
    ap.add_argument("--nc-dir", default=None, help="(with --daily) write every tile of this rank into the reference's NetCDF-4 tile files "
                                                   "<dir>/<tile_id>/<tile_id>_<var>.nc while the next one runs (ncio.TileSink)")
    ap.add_argument("--precision", default="auto", choices=("auto", "fast", "exact"), help="--nc-dir / --tile-dir: PrecisionPolicy of the streamed run")
    ap.add_argument("--chunk", type=int, default=50, help="chunk edge of the tile files (tiling.py:453-486: 50)")
    ap.add_argument("--deflate", action="store_true", help="--nc-dir: store the daily variables with shuffle + deflate, the chunk bytes "
                                                           "formed on the GPU (twx_stream_deflate)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from . import _lib, synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    grid, tmin, tmax = synth.make_case(args.config, with_obs=args.daily)
    even = grid["mask"].shape[0] % args.tile == 0 and grid["mask"].shape[1] % args.tile == 0
    if args.nc_dir and not args.daily:
        raise SystemExit("--nc-dir writes the daily tile files: add --daily")
    if (args.tile_dir or args.nc_dir) and not even:
        # the streamed writer works on tiles of ONE shape (twx_stream_*): say so before anything is computed
        raise SystemExit("--tile-dir needs a grid that --tile divides evenly (grid %dx%d, tile %d): choose another "
                         "--tile or drop --tile-dir" % (grid["mask"].shape + (args.tile,)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    ctx = _lib.Context(device=local)
    ctx.set_stations(_lib.TMIN, tmin, with_obs=args.daily)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=args.daily)
    tiles = tile_list(grid["mask"], args.tile, args.tile)
    assignment = assign_tiles(tiles, world)
    t0 = time.perf_counter()
    if args.nc_dir:
        # streamed into NetCDF-4 tile files; --deflate: the GPU hands over chunk bytes, the sink appends them
        from . import ncio
        from .interp import Tiler
        if args.tile % args.chunk:
            raise SystemExit("--chunk must divide --tile")
        info = Tiler(grid, args.tile, args.tile, args.chunk, args.chunk, process_tiles=()).build_tile_grid_info()
        sink = ncio.TileSink(info, args.nc_dir, tmin.days, zlib=args.deflate, order=[t[0] for t in assignment[rank]])
        _, _, dev_ms = interp_tiles_streamed(ctx, grid, assignment[rank], args.tile, args.tile, daily=True, sink=sink, precision=args.precision,
                                             deflate_chunks=(args.chunk, args.chunk) if args.deflate else None)
        sink.close()
        mosaic = None
    elif args.tile_dir:
        # streamed: outputs of tile t travel to the host and to disk while tile t + 1 is computed; the normals (small)
        # are kept for --gather
        os.makedirs(args.tile_dir, exist_ok=True)
        mine = {}

        def sink(k, arrays):
            np.savez(os.path.join(args.tile_dir, "tile%05d.npz" % k), **{n: v for n, v in arrays.items() if hasattr(v, "shape")})
            if args.gather:
                mine[k] = {n: np.array(arrays[n]) for n in NORMAL_KEYS}
        _, _, dev_ms = interp_tiles_streamed(ctx, grid, assignment[rank], args.tile, args.tile, daily=args.daily, sink=sink, precision=args.precision)
        mosaic = gather_mosaic(mine, assignment, grid["mask"].shape, args.tile, args.tile, NORMAL_KEYS, rank, world,
                               device="cuda:%d" % local) if args.gather else None
    elif args.daily:
        mine = interp_tiles(grid, gpu_compute(ctx, daily=True), assignment[rank], args.tile, args.tile)
        dev_ms = None
        mosaic = gather_mosaic(mine, assignment, grid["mask"].shape, args.tile, args.tile, NORMAL_KEYS, rank, world,
                               device="cuda:%d" % local) if args.gather else None
    else:
        # normals only: everything stays in HBM, tiles land in the gather's send buffer
        dgrid = upload_grid(grid, "cuda:%d" % local, assignment[rank], args.tile, args.tile)   # this rank's tiles only
        nmax = max(len(a) for a in assignment)
        buf, _, ms = interp_tiles_device(ctx, dgrid, assignment[rank], args.tile, args.tile, nslots=nmax)
        dev_ms = float(sum(ms))
        mosaic = None
        if args.gather:
            mosaic = gather_mosaic_device(buf, assignment, grid["mask"].shape, args.tile, args.tile, rank, world)
            if mosaic is not None:
                mosaic = {k: v.cpu().numpy() for k, v in mosaic.items()}
    dt = time.perf_counter() - t0
    ncell = sum(t[3] for t in assignment[rank])
    print(json.dumps({"rank": rank, "tiles": len(assignment[rank]), "cells": ncell, "seconds": dt, "device_ms": dev_ms}), flush=True)
    if rank == 0 and args.out and mosaic is not None:
        np.savez_compressed(args.out, **mosaic)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
