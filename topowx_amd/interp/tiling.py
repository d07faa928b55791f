"""Chunk feed of the interpolation grid (twx/interp/tiling.py:44-302).

In-memory counterpart of the reference's ``Tiler``: cuts the grid into tiles and
work chunks and yields the f8[5+N, Y, X] work chunk in the reference's plane order.
netCDF tile writing is SURVEY.md 8f-2; mosaicking / aggregation live in ``aggregate.py`` (8f-3).
"""
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
    """``grid`` is the dict of topowx_amd.synth.make_grid (mask, lat, lon, planes)."""

    ATTRS = ("elev", "tdi", "climdiv") + tuple(("lst_night", m) for m in range(12)) + \
        tuple(("lst_day", m) for m in range(12))

    def __init__(self, grid, tile_size_y, tile_size_x, chk_size_y, chk_size_x, process_tiles=None):
        self.grid = grid
        self.mask = np.asarray(grid["mask"], bool)
        self.lons, self.lats = grid["lon"], grid["lat"]
        self.nrows, self.ncols = self.mask.shape
        if self.nrows % tile_size_y or self.ncols % tile_size_x or tile_size_y % chk_size_y or tile_size_x % chk_size_x:
            raise ValueError("grid / tile / chunk sizes must divide evenly (tiling.py:62-74)")
        self.tile_size_y, self.tile_size_x = tile_size_y, tile_size_x
        self.chk_size_y, self.chk_size_x = chk_size_y, chk_size_x
        self.chk_size_i = 5 + len(self.ATTRS)
        self.tile_ids, self.tile_rc = {}, {}
        self.tile_chks = []
        k = 0
        for cy, i in enumerate(range(0, self.nrows, tile_size_y)):
            for cx, j in enumerate(range(0, self.ncols, tile_size_x)):
                if not self.mask[i:i + tile_size_y, j:j + tile_size_x].any():
                    continue                         # tiles without a valid cell get no number (tiling.py:141)
                name = "h%02dv%02d" % (cx, cy)
                self.tile_ids[k] = name
                self.tile_rc[name] = (i, j)
                if process_tiles is None or k in process_tiles:
                    for y in range(0, tile_size_y, chk_size_y):
                        for x in range(0, tile_size_x, chk_size_x):
                            self.tile_chks.append((k, i, j, y, x))
                k += 1
        self.ntiles = len({c[0] for c in self.tile_chks})
        self.ntile_chks = len(self.tile_chks)
        self.iter_x = 0

    def __iter__(self):
        return self

    def _plane(self, a, i, j, y, x):
        if isinstance(a, tuple):
            arr = self.grid[a[0]][a[1]]
        else:
            arr = self.grid[a]
        return arr[i + y:i + y + self.chk_size_y, j + x:j + x + self.chk_size_x]

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
        for z, a in enumerate(self.ATTRS):
            w[5 + z] = self._plane(a, i, j, y, x)
        return k, w

    next = __next__

    def build_tile_grid_info(self):
        return TileGridInfo(self.tile_ids, self.tile_rc, self.ntiles, self.lons, self.lats, self.tile_size_y,
                            self.tile_size_x, self.chk_size_y, self.chk_size_x, self.chk_size_i)
