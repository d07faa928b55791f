"""Work-chunk level counterpart of ``scripts/step25_mpi_interp_tair.py`` (proc_work, :49-198).

Keeps the reference's structure -- ``Tiler`` yields f8[32, Y, X] work chunks, a ``PtInterpTair`` interpolates them,
a writer stores ``days x Y x X`` int16 / ``12 x Y x X`` f4 blocks per tile -- but one library call replaces the
2 500-iteration Python cell loop, ranks own whole tiles (no per-tile write token, step25:177-196) and the chunks run
through a ``TileStream`` (``twx_stream_*``): while the GPU interpolates chunk i + 1, the outputs of chunk i travel
to pinned host memory and are written into their tile; a finished tile goes to disk and is dropped at once
(the reference writes every chunk as it is produced, step25:177-185).

* resume: with ``check_tiles_done`` a tile whose output already exists in ``out_dir`` is skipped, as the
  reference's ``Tiler.get_incomplete_tile_nums`` does from the directory listing (tiling.py:111-122,258-275,
  step25:354-356).  Outputs are written under a temporary name and renamed when the tile is complete, so an
  interrupted tile is redone.
* log: one JSON line per tile (cells, cells ok, failures by TWX_CELL_* code, device ms, output bytes, seconds).

Tiles are written as ``<tile_id>.npz`` or, with ``out_format="nc"``, through ``ncio.TileWriter`` as
``<tile_id>/<tile_id>_<var>.nc``.
"""
import json
import os
import shutil
import time

import numpy as np

from . import _lib
from .driver import PrecisionPolicy, assign_tiles, tile_list
from .interp import PtInterpTair, Tiler
from .interp.interp_tair import chunk_to_grid

__all__ = ["TileStore", "proc_work", "tiles_done"]


class TileStore(object):
    """In-memory stand-in of ``TileWriter`` (tiling.py:304-537): result arrays of one tile, pre-filled
    with the netCDF fill values (step25:68-88), chunks written at (str_row, str_col)."""

    def __init__(self, ndays, tile_y, tile_x, daily):
        self.a = {}
        for v in ("tmin", "tmax"):
            self.a["norm_" + v] = np.full((12, tile_y, tile_x), _lib.FILL_F4, np.float32)
            self.a["se_" + v] = np.full((12, tile_y, tile_x), _lib.FILL_F4, np.float32)
            if daily:
                self.a["daily_" + v] = np.full((ndays, tile_y, tile_x), _lib.FILL_I2, np.int16)
        self.a["ninvalid"] = np.full((tile_y, tile_x), _lib.FILL_I4, np.int32)
        self.a["status"] = np.full((tile_y, tile_x), -1, np.int32)

    def write_tile_chunk(self, str_row, str_col, out):
        for k, v in out.items():
            if k in self.a:
                y, x = v.shape[-2:]
                self.a[k][..., str_row:str_row + y, str_col:str_col + x] = v

    def nbytes(self):
        return int(sum(v.nbytes for v in self.a.values()))

    def save(self, path):
        np.savez_compressed(path, **self.a)


def tiles_done(out_dir, tile_ids):
    """Names of the tiles whose output exists in ``out_dir`` (tiling.py:258-275: the directory listing decides)."""
    if out_dir is None or not os.path.isdir(out_dir):
        return set()
    names = set(os.listdir(out_dir))
    return {t for t in tile_ids if t in names or (t + ".npz") in names}


# out_format -> netCDF container: "nc" = the reference's NetCDF-4 where libhdf5 is loadable (else classic netCDF)
NC_FORMATS = {"nc": None, "nc4": "NETCDF4", "nc3": "NETCDF3_64BIT"}


def _write_tile(out_dir, out_format, info, tile_id, store, days):
    """Write under a temporary name, rename when complete (an interrupted tile is not mistaken for a finished one)."""
    if out_format in NC_FORMATS:                                               # step25:181-185
        from .ncio import TileWriter
        tmp = os.path.join(out_dir, tile_id + ".part")
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        # files go to <tmp>/<tile_id>/<tile_id>_<var>.nc, the directory is moved when complete
        writer = TileWriter(info, tmp, format=NC_FORMATS[out_format])
        for v in ("tmin", "tmax"):
            writer.write_tile_chunk(tile_id, v, days, 0, 0, store.a.get("daily_" + v), store.a["norm_" + v],
                                    store.a["se_" + v], store.a["ninvalid"])
        final = os.path.join(out_dir, tile_id)
        shutil.rmtree(final, ignore_errors=True)
        os.replace(os.path.join(tmp, tile_id), final)
        shutil.rmtree(tmp, ignore_errors=True)
    else:
        tmp = os.path.join(out_dir, tile_id + ".part.npz")
        store.save(tmp)
        os.replace(tmp, os.path.join(out_dir, tile_id + ".npz"))


def proc_work(grid, stn_da_tmin, stn_da_tmax, tile_size=250, chunk_size=50, daily=True, out_dir=None,
              rank=0, world=1, device=0, out_format="npz", check_tiles_done=True, keep=None, log=None, precision="fast"):
    """Interpolate the tiles of this rank chunk by chunk.

    ``precision``: "fast" (default) | "exact" | "auto" (``driver.PrecisionPolicy``): exact = every kriging system on the fp64
    covariance build -- outputs equal an fp64 evaluation to the last int16 / f4 bit; auto = exact for as long as the chunks'
    kernels hide behind their copy-out to the host, else fast.  The mode a tile's last chunk ran in is part of its log record;
    a run that was not asked for "fast" ends with one more record, the policy's summary (``{"requested": ..., "precision": ...}``).
    (The default stays "fast" here, unlike ``driver.interp_tiles_streamed``: 50 x 50 chunks are too small to hide their kernels
    behind their copy-out, and a resumed run should redo a tile in the mode it was first written in.)

    Returns ``{tile_id: TileStore}`` of the tiles kept in memory (``keep``; default: only when nothing is written
    to ``out_dir``).  ``log``: a callable or file object receiving one JSON line per tile (default: none); the
    records are also returned as ``proc_work.last_log``."""
    keep = (out_dir is None) if keep is None else keep
    tiles = tile_list(grid["mask"], tile_size, tile_size)
    mine = {t[0] for t in assign_tiles(tiles, world)[rank]}
    probe = Tiler(grid, tile_size, tile_size, chunk_size, chunk_size, process_tiles=())
    records = []
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        if check_tiles_done:                                                   # step25:354-356
            done = tiles_done(out_dir, [probe.tile_ids[k] for k in mine])
            for k in sorted(mine):
                if probe.tile_ids[k] in done:
                    records.append({"tile": probe.tile_ids[k], "skipped": True, "rank": rank})
            mine = {k for k in mine if probe.tile_ids[k] not in done}
    tiler = Tiler(grid, tile_size, tile_size, chunk_size, chunk_size, process_tiles=mine)
    info = tiler.build_tile_grid_info()
    pt_interp = PtInterpTair(stn_da_tmin, stn_da_tmax, norms_only=not daily, device=device)   # step25:53-55
    days = pt_interp.days
    stream = pt_interp.ctx.stream(chunk_size, chunk_size, daily=daily, nslots=2)
    policy = PrecisionPolicy(pt_interp.ctx, precision)
    stores, open_tiles, pending = {}, {}, None

    def emit(rec):
        records.append(rec)
        if log is not None:
            line = json.dumps(rec)
            log(line) if callable(log) else log.write(line + "\n")

    def finish(p):
        slot, tile_id, str_row, str_col, mode = p
        out = stream.wait(slot)                                                    # outputs of that chunk are on the host
        policy.observe(mode, out["device_ms"], stream.times(slot)[1])
        st = open_tiles[tile_id]
        st["store"].write_tile_chunk(str_row, str_col, out)                        # step25:181-185
        st["device_ms"] += out["device_ms"]
        st["left"] -= 1
        if st["left"] == 0:                                                        # last chunk: write, log, drop
            store = st["store"]
            status = store.a["status"]
            codes, counts = np.unique(status[status > 0], return_counts=True)
            if out_dir is not None:
                _write_tile(out_dir, out_format, info, tile_id, store, days)
            emit({"tile": tile_id, "rank": rank, "cells": int((status != -1).sum()), "ok": int((status == 0).sum()),
                  "failures": {int(c): int(n) for c, n in zip(codes, counts)}, "device_ms": round(st["device_ms"], 3),
                  "bytes": store.nbytes(), "seconds": round(time.perf_counter() - st["t0"], 3), "precision": mode})
            if keep:
                stores[tile_id] = store
            del open_tiles[tile_id]

    try:
        for n, (tile_num, wrk_chk) in enumerate(tiler):                              # step25:94-96
            tile_id = info.get_tile_id(tile_num)
            if tile_id not in open_tiles:
                open_tiles[tile_id] = {"store": TileStore(days.size, tile_size, tile_size, daily), "left": info.chks_per_tile,
                                       "device_ms": 0.0, "t0": time.perf_counter()}
            str_row, str_col = int(wrk_chk[0, 0, 0]), int(wrk_chk[1, 0, 0])          # step25:110-111
            slot = n & 1
            mode = policy.mode
            stream.submit(slot, chunk_to_grid(wrk_chk))                              # step25:126-172 in one call, asynchronous
            if pending is not None:
                finish(pending)                                                      # overlaps the chunk just submitted
            pending = (slot, tile_id, str_row, str_col, mode)
        if pending is not None:
            finish(pending)
    finally:
        stream.close()
        policy.close()
        pt_interp.close()
    if precision != "fast":
        emit(dict(policy.summary(), rank=rank))
    proc_work.last_log = records
    return stores


proc_work.last_log = []
