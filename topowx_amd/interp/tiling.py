"""Chunk feed of the interpolation grid (twx/interp/tiling.py:44-302).

Counterpart of the reference's ``Tiler``: cuts the grid into tiles and work chunks and yields the f8[5+N, Y, X] work
chunk in the reference's plane order.  Two constructor shapes: the reference's (``Tiler(ds_mask, ds_attr_ls, tile_size_y,
tile_size_x, chk_size_y, chk_size_x, path_out, process_tiles)`` on netCDF datasets -- ``topowx_amd.ncio.open_dataset`` /
``topowx_amd.h5nc.Dataset`` objects, read by hyperslab per chunk, step25:266-289) and the in-memory form on the dict of
``topowx_amd.synth.make_grid``.  The netCDF tile writer is ``topowx_amd.ncio.TileWriter``;
mosaicking / aggregation live in ``aggregate.py`` (8f-3).
"""
import os

import numpy as np

__all__ = ["Tiler", "TileGridInfo"]


class TileGridInfo(object):
    def __init__(self, tile_ids, tile_rc, ntiles, lons, lats, tile_size_y, tile_size_x, chk_size_y, chk_size_x,
                 chk_size_i):
        self.tile_ids, self.tile_rc, self.ntiles = tile_ids, tile_rc, ntiles
        self.lons, self.lats = lons, lats
        self.tile_size_y, self.tile_size_x = tile_size_y, tile_size_x
        self.chk_size_y, self.chk_size_x, self.chk_size_i = chk_size_y, chk_size_x, chk_size_i
        self.chks_per_tile = (tile_size_x // chk_size_x) * (tile_size_y // chk_size_y)
        self.nchks = self.chks_per_tile * ntiles

    def get_tile_id(self, tile_num):
        return self.tile_ids[tile_num]


class Tiler(object):
    """``Tiler(grid, tile_size_y, tile_size_x, chk_size_y, chk_size_x, process_tiles=None)`` on the dict of
    topowx_amd.synth.make_grid (mask, lat, lon, planes), or the reference's
    ``Tiler(ds_mask, ds_attr_ls, tile_size_y, tile_size_x, chk_size_y, chk_size_x, path_out, process_tiles=False)``
    (tiling.py:50-128): ``ds_mask`` holds ``mask`` / ``lon`` / ``lat``, ``ds_attr_ls`` is a list of
    ``(variable name, dataset)`` -- the planes 5.. of the work chunk in list order; ``process_tiles`` a list of tile
    numbers, or True = only tiles without a directory under ``path_out`` (tiling.py:258-275)."""

    ATTRS = ("elev", "tdi", "climdiv") + tuple(("lst_night", m) for m in range(12)) + \
        tuple(("lst_day", m) for m in range(12))

    def __init__(self, grid, *args, **kwargs):
        if isinstance(grid, dict):
            names = ("tile_size_y", "tile_size_x", "chk_size_y", "chk_size_x", "process_tiles")
            a = dict(zip(names, args))
            a.update(kwargs)
            process_tiles = a.get("process_tiles")
            self.grid = grid
            self.attrs = None
            self.mask = np.asarray(grid["mask"], bool)
            self.lons, self.lats = grid["lon"], grid["lat"]
            nattr = len(self.ATTRS)
        else:
            names = ("ds_attr_ls", "tile_size_y", "tile_size_x", "chk_size_y", "chk_size_x", "path_out", "process_tiles")
            a = dict(zip(names, args))
            a.update(kwargs)
            ds_mask = grid
            self.grid = None
            self.mask = np.array(ds_mask.variables["mask"][:], dtype=bool)
            self.lons = np.asarray(ds_mask.variables["lon"][:], np.float64)
            self.lats = np.asarray(ds_mask.variables["lat"][:], np.float64)
            self.attrs = [ds.variables[varname] for varname, ds in a["ds_attr_ls"]]   # raw values: no mask / scale (:101-103)
            nattr = len(self.attrs)
            process_tiles = a.get("process_tiles", False)
        tile_size_y, tile_size_x = a["tile_size_y"], a["tile_size_x"]
        chk_size_y, chk_size_x = a["chk_size_y"], a["chk_size_x"]
        self.nrows, self.ncols = self.mask.shape
        if self.nrows % tile_size_y or self.ncols % tile_size_x or tile_size_y % chk_size_y or tile_size_x % chk_size_x:
            raise ValueError("grid / tile / chunk sizes must divide evenly (tiling.py:62-74)")
        self.tile_size_y, self.tile_size_x = tile_size_y, tile_size_x
        self.chk_size_y, self.chk_size_x = chk_size_y, chk_size_x
        self.chk_size_i = 5 + nattr
        self.tile_ids, self.tile_rc = {}, {}
        tiles = []
        k = 0
        for cy, i in enumerate(range(0, self.nrows, tile_size_y)):
            for cx, j in enumerate(range(0, self.ncols, tile_size_x)):
                if not self.mask[i:i + tile_size_y, j:j + tile_size_x].any():
                    continue                         # tiles without a valid cell get no number (tiling.py:141)
                name = "h%02dv%02d" % (cx, cy)
                self.tile_ids[k] = name
                self.tile_rc[name] = (i, j)
                tiles.append((k, i, j))
                k += 1
        if isinstance(process_tiles, bool):
            process_tiles = self.get_incomplete_tile_nums(a["path_out"]) if process_tiles else None
        elif process_tiles is not None:
            process_tiles = list(process_tiles)
        self.process_tiles = process_tiles
        self.tile_chks = []
        for k, i, j in tiles:
            if process_tiles is None or k in process_tiles:
                for y in range(0, tile_size_y, chk_size_y):
                    for x in range(0, tile_size_x, chk_size_x):
                        self.tile_chks.append((k, i, j, y, x))
        self.ntiles = len({c[0] for c in self.tile_chks})
        self.ntile_chks = len(self.tile_chks)
        self.iter_x = 0

    def get_incomplete_tile_nums(self, path_out):
        """Tile numbers without a directory under ``path_out`` (tiling.py:258-275: only the directory's existence is
        checked, not whether the tile is complete)."""
        name_to_id = {name: k for k, name in self.tile_ids.items()}
        done = {name_to_id[n] for n in os.listdir(path_out) if n in name_to_id}
        return np.array(sorted(set(self.tile_ids) - done), dtype=np.int64)

    def __iter__(self):
        return self

    def _plane(self, z, i, j, y, x):
        rs = slice(i + y, i + y + self.chk_size_y)
        cs = slice(j + x, j + x + self.chk_size_x)
        if self.attrs is not None:
            return self.attrs[z][rs, cs]
        a = self.ATTRS[z]
        arr = self.grid[a[0]][a[1]] if isinstance(a, tuple) else self.grid[a]
        return arr[rs, cs]

    def __next__(self):
        if self.iter_x == self.ntile_chks:
            raise StopIteration()
        k, i, j, y, x = self.tile_chks[self.iter_x]
        self.iter_x += 1
        w = np.full((self.chk_size_i, self.chk_size_y, self.chk_size_x), np.nan)
        rr, cc = np.mgrid[y:y + self.chk_size_y, x:x + self.chk_size_x]
        w[0], w[1] = rr, cc
        w[2] = self.mask[i + y:i + y + self.chk_size_y, j + x:j + x + self.chk_size_x]
        w[3] = self.lats[i + y:i + y + self.chk_size_y][:, None]
        w[4] = self.lons[j + x:j + x + self.chk_size_x][None, :]
        for z in range(self.chk_size_i - 5):
            w[5 + z] = self._plane(z, i, j, y, x)
        return k, w

    next = __next__

    def build_tile_grid_info(self):
        return TileGridInfo(self.tile_ids, self.tile_rc, self.ntiles, self.lons, self.lats, self.tile_size_y,
                            self.tile_size_x, self.chk_size_y, self.chk_size_x, self.chk_size_i)
