"""Work-chunk level counterpart of ``scripts/step25_mpi_interp_tair.py`` (proc_work, :49-198).

Keeps the reference's structure -- ``Tiler`` yields f8[32, Y, X] work chunks, a ``PtInterpTair``
interpolates them, a writer stores ``days x Y x X`` int16 / ``12 x Y x X`` f4 blocks per tile -- but
one ``interp_chunk`` call replaces the 2 500-iteration Python cell loop, and ranks own whole tiles
(no per-tile write token, step25:177-196).  Tiles are written as ``<tile_id>.npz`` or, with
``out_format="nc"``, through ``ncio.TileWriter`` as ``<tile_id>/<tile_id>_<var>.nc``.
"""
import os

import numpy as np

from . import _lib
from .driver import assign_tiles, tile_list
from .interp import PtInterpTair, Tiler

__all__ = ["TileStore", "proc_work"]


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

    def save(self, path):
        np.savez_compressed(path, **self.a)


def proc_work(grid, stn_da_tmin, stn_da_tmax, tile_size=250, chunk_size=50, daily=True, out_dir=None,
              rank=0, world=1, device=0, out_format="npz"):
    """Interpolate the tiles of this rank chunk by chunk; returns {tile_id: TileStore}."""
    tiles = tile_list(grid["mask"], tile_size, tile_size)
    mine = {t[0] for t in assign_tiles(tiles, world)[rank]}
    tiler = Tiler(grid, tile_size, tile_size, chunk_size, chunk_size, process_tiles=mine)
    info = tiler.build_tile_grid_info()
    pt_interp = PtInterpTair(stn_da_tmin, stn_da_tmax, norms_only=not daily, device=device)   # step25:53-55
    stores = {}
    for tile_num, wrk_chk in tiler:                                   # step25:94-96
        tile_id = info.get_tile_id(tile_num)
        store = stores.setdefault(tile_id, TileStore(pt_interp.days.size, tile_size, tile_size, daily))
        str_row, str_col = int(wrk_chk[0, 0, 0]), int(wrk_chk[1, 0, 0])            # step25:110-111
        out = pt_interp.interp_chunk(wrk_chk, daily=daily)                         # step25:126-172 in one call
        store.write_tile_chunk(str_row, str_col, out)                              # step25:181-185
    days = pt_interp.days
    pt_interp.close()
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        if out_format == "nc":                                                     # step25:181-185
            from .ncio import TileWriter
            writer = TileWriter(info, out_dir)
            for tile_id, store in stores.items():
                for v in ("tmin", "tmax"):
                    writer.write_tile_chunk(tile_id, v, days, 0, 0, store.a.get("daily_" + v), store.a["norm_" + v],
                                            store.a["se_" + v], store.a["ninvalid"])
        else:
            for tile_id, store in stores.items():
                store.save(os.path.join(out_dir, tile_id + ".npz"))
    return stores
