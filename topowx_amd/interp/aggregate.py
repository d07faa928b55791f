"""Mosaicking and monthly / annual aggregation of the tile product.

Counterparts of ``twx/interp/tiling.py``: ``TileMosaic`` (:553-971), ``_TairAggregate`` (:1080-1166) and
``write_ds_mthly`` (:1169-1219, scripts/step27_create_monthly.py), at two levels: on arrays (tile stores in memory)
and on files with the reference's signatures (``TileMosaic(fpath_mask, ...).create_dly_ann_mosaics(tiles, varname,
path_in, path_out, start_yr, end_yr, ds_version_str, chunk_cache_size)``, ``create_normals_mosaic(tiles, varname,
path_in, fpath_out, ds_version_str)``, ``write_ds_mthly(ds_dly, fpath_out, varname, yr, ds_version_str)``; the
containers are ``topowx_amd.ncio``).  The arithmetic between the files -- the means over
(year, month) groups, the rounding and the int16 packing -- runs in libtwxhip (``twx_aggregate``, ``twx_pack_i16``).
"""
import os

import numpy as np

from .. import _lib
from ..dates import MONTH, YEAR, get_mth_metadata

__all__ = ["TairAggregate", "TileMosaic", "mthly_from_daily", "write_ds_mthly"]


class TairAggregate(object):
    """``_TairAggregate`` (tiling.py:1080-1166): first axis of every input is time."""

    def __init__(self, days, device=0):
        self.days = days
        u_yrs, u_mths = np.unique(days[YEAR]), np.unique(days[MONTH])
        self.u_yrs = u_yrs
        yr_mths = get_mth_metadata(int(u_yrs[0]), int(u_yrs[-1]))
        # tiling.py:1110-1111 keeps every year between the first and the last; the day groups only exist
        # for years that occur (tiling.py:1101), so the two agree only for a gap-free day axis
        if u_yrs.size != int(u_yrs[-1]) - int(u_yrs[0]) + 1:
            raise ValueError("day axis skips whole years")
        self.yr_mths = yr_mths[np.isin(yr_mths[MONTH], u_mths)]
        self._ctx = _lib.Context(device)
        self._ctx.set_days(days)
        self._ctx_mth = None
        self.nyr, self.nmth = self._ctx.aggregate_dims()

    def close(self):
        for c in (self._ctx, self._ctx_mth):
            if c is not None:
                c.close()
        self._ctx = self._ctx_mth = None

    @staticmethod
    def _dense(tair):
        """masked / plain array -> contiguous int16 / f4 / f8 with NaN at masked cells."""
        if np.ma.isMaskedArray(tair):
            a = np.ma.getdata(tair)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            return np.where(np.ma.getmaskarray(tair), np.nan, a)
        a = np.asarray(tair)
        if a.dtype not in (np.int16, np.float32, np.float64):
            a = a.astype(np.float64)
        return a

    def daily_to_mthly(self, tair):
        out = self._ctx.aggregate(self._dense(tair), mthly=True)["mthly"]
        return np.ma.masked_invalid(out)

    def daily_to_ann(self, tair):
        out = self._ctx.aggregate(self._dense(tair), mthly=False, ann=True)["ann"]
        return np.ma.masked_invalid(out)

    def mthly_to_ann(self, tair_mthly):
        """Per year the mean of its monthly values (tiling.py:1151-1166): the same group-mean kernel on
        the month axis, one group per year."""
        if self._ctx_mth is None:
            self._ctx_mth = _lib.Context(self._ctx.device)
            self._ctx_mth.set_days({MONTH: np.ones(self.yr_mths.size, np.int32), YEAR: self.yr_mths[YEAR]})
        a = self._dense(tair_mthly)
        if a.dtype == np.int16:
            a = a.astype(np.float64)
        out = self._ctx_mth.aggregate(a, mthly=True)["mthly"]
        return np.ma.masked_invalid(out)

    def daily_i16_to_mthly_i16(self, daily_raw):
        """write_ds_mthly (tiling.py:1169-1219) on the raw 'i2' product: unpack (scale 0.01, fill
        -32767), group means, ``np.ma.round(., 2)``, pack with the same scale -> int16 [nyr*nmth, ...]."""
        daily_raw = np.ascontiguousarray(daily_raw)
        if daily_raw.dtype != np.int16:
            raise TypeError("daily_i16_to_mthly_i16 takes the raw int16 product")
        return self._ctx.aggregate(daily_raw, mthly=False, mthly_i16=True)["mthly_i16"]


def mthly_from_daily(daily_raw, days, device=0):
    """One-call form of write_ds_mthly's arithmetic for one year (or more) of the daily mosaic."""
    agg = TairAggregate(days, device=device)
    try:
        return agg.daily_i16_to_mthly_i16(daily_raw)
    finally:
        agg.close()


def write_ds_mthly(ds_dly, fpath_out, varname, yr, ds_version_str, format=None, device=0):
    """``write_ds_mthly`` (tiling.py:1169-1219, step27): the monthly file of one year's daily mosaic ``ds_dly`` (an open
    ``ncio.open_dataset`` / ``h5nc.Dataset``).  The daily variable is read in bands of its chunk rows, unpacked as
    netCDF4-python unpacks it, averaged per (year, month), rounded to 2 decimals and packed with the variable's scale
    on the GPU (``twx_aggregate``: bit-exact against the executed reference slice, tests/test_gpu_agg.py)."""
    from .. import ncio
    ds_out = ncio.create_ds_mthly(ds_dly, fpath_out, yr, varname, ds_version_str, format=format)
    try:
        days = ncio.days_of(ds_dly)
        var = ds_dly.variables[varname]
        chk = var.chunking()
        nrow, ncol = var.shape[1], var.shape[2]
        step = max(int(chk[1]) if chk != "contiguous" else nrow, 1)
        # bands of whole chunk rows, a few hundred MB of int16 at a time
        step *= max(1, (256 << 20) // max(1, 2 * var.shape[0] * ncol * step))
        agg = TairAggregate(days, device=device)
        try:
            out = np.empty((agg.nyr * agg.nmth, nrow, ncol), np.int16)
            for r in range(0, nrow, step):
                out[:, r:r + step, :] = agg.daily_i16_to_mthly_i16(np.ascontiguousarray(var[:, r:r + step, :], np.int16))
        finally:
            agg.close()
        if out.shape[0] != 12:
            raise ValueError("write_ds_mthly takes ONE year of daily data (tiling.py:1176-1178)")
        ds_out.variables[varname][:] = out
        ds_out.sync()
    finally:
        ds_out.close()
    return fpath_out


class TileMosaic(object):
    """Assemble tile results into mosaics (tiling.py:553-971).  The mosaic spans the tile rows / columns between the
    first and the last tile given (tiling.py:570-596); tiles that are missing stay at the fill value (:772-776).
    ``TileMosaic(tile_grid_info)`` works on ``step25.TileStore`` objects in memory; the reference's
    ``TileMosaic(fpath_mask, tile_size_y, tile_size_x, chk_size_y, chk_size_x)`` (tiling.py:557-565) reads the tile
    layout from the mask file and serves the file-level methods."""

    def __init__(self, tile_grid_info, tile_size_y=None, tile_size_x=None, chk_size_y=None, chk_size_x=None, device=0):
        if isinstance(tile_grid_info, (str, os.PathLike)):
            from .. import ncio
            from .tiling import Tiler
            ds_mask = ncio.open_dataset(os.fspath(tile_grid_info))
            try:
                self.atiler = Tiler(ds_mask, [], tile_size_y, tile_size_x, chk_size_y, chk_size_x, None)
            finally:
                ds_mask.close()
            tile_grid_info = self.atiler.build_tile_grid_info()
            self.lon, self.lat = self.atiler.lons, self.atiler.lats
        self.tinfo = tile_grid_info
        self.device = device

    # ---- file level (tiling.py:567-780, :782-971) -------------------------------------------------------------------
    def _mosaic_grid(self, tiles):
        """Tile names of the bounding tile rectangle, its first row / column in the grid, lon / lat of the mosaic."""
        tcols = [int(t[1:3]) for t in tiles]
        trows = [int(t[4:]) for t in tiles]
        tcols = list(range(min(tcols), max(tcols) + 1))
        trows = list(range(min(trows), max(trows) + 1))
        ty, tx = self.tinfo.tile_size_y, self.tinfo.tile_size_x
        rc = [self.tinfo.tile_rc[n] for n in ("h%02dv%02d" % (c, r) for c in tcols for r in trows) if n in self.tinfo.tile_rc]
        if not rc:
            raise ValueError("none of the tiles exists in the grid")
        min_row, min_col = min(r for r, _ in rc), min(c for _, c in rc)
        max_row, max_col = max(r for r, _ in rc), max(c for _, c in rc)
        lon = np.asarray(self.tinfo.lons)[min_col:max_col + tx]
        lat = np.asarray(self.tinfo.lats)[min_row:max_row + ty]
        return tcols, trows, lon, lat

    @staticmethod
    def _tile_path(path_in, name, varname):
        return os.path.join(path_in, name, "%s_%s.nc" % (name, varname))

    def create_dly_ann_mosaics(self, tiles, varname, path_in, path_out, start_yr, end_yr, ds_version_str,
                               chunk_cache_size=None, format=None):
        """``create_dly_ann_mosaics`` (tiling.py:567-780): one ``<varname>_<year>.nc`` per year of ``start_yr..end_yr``
        under ``path_out`` holding the raw int16 daily values of every tile file found under ``path_in`` (a missing
        tile stays at the fill value).  Returns the paths."""
        from .. import ncio
        tcols, trows, lon, lat = self._mosaic_grid(tiles)
        ty, tx = self.tinfo.tile_size_y, self.tinfo.tile_size_x
        names = [("h%02dv%02d" % (c, r), i, j) for i, c in enumerate(tcols) for j, r in enumerate(trows)]
        first = next((n for n, _, _ in names if os.path.exists(self._tile_path(path_in, n, varname))), None)
        if first is None:
            raise IOError("no tile file of %s under %s" % (varname, path_in))
        ds_tile = ncio.open_dataset(self._tile_path(path_in, first, varname))
        days_all = ncio.days_of(ds_tile)
        ds_tile.close()
        keep = np.nonzero((days_all[YEAR] >= start_yr) & (days_all[YEAR] <= end_yr))[0]
        yrs = np.unique(days_all[YEAR][keep])
        yr_rows = [keep[days_all[YEAR][keep] == yr] for yr in yrs]
        os.makedirs(path_out, exist_ok=True)
        paths = [os.path.join(path_out, "%s_%d.nc" % (varname, yr)) for yr in yrs]
        yr_ds = [ncio.create_dly_mosaic_ds(p, varname, days_all[rows], lon, lat, ds_version_str, format=format)
                 for p, rows in zip(paths, yr_rows)]
        try:
            for name, i, j in names:
                fp = self._tile_path(path_in, name, varname)
                if not os.path.exists(fp):
                    continue                               # "Tile does not exist. Values for tile will be fill values."
                ds_tile = ncio.open_dataset(fp, rdcc_nbytes=chunk_cache_size)
                try:
                    v = ds_tile.variables[varname]
                    for ds, rows in zip(yr_ds, yr_rows):   # a year's days are contiguous on a gap-free axis
                        ds.variables[varname][:, j * ty:(j + 1) * ty, i * tx:(i + 1) * tx] = v[rows[0]:rows[-1] + 1, :, :]
                finally:
                    ds_tile.close()
        finally:
            for ds in yr_ds:
                ds.close()
        return paths

    def _normals_mosaic_file(self, tiles, varname, path_in, fpath_out, ds_version_str, format=None):
        """``create_normals_mosaic`` (tiling.py:782-971) on files: packed int16 normals / SE of every tile found."""
        from .. import ncio
        tcols, trows, lon, lat = self._mosaic_grid(tiles)
        ty, tx = self.tinfo.tile_size_y, self.tinfo.tile_size_x
        ds = ncio.create_normals_mosaic_ds(fpath_out, varname, lon, lat, ds_version_str, format=format)
        ctx = _lib.Context(self.device)
        try:
            for i, c in enumerate(tcols):
                for j, r in enumerate(trows):
                    fp = self._tile_path(path_in, "h%02dv%02d" % (c, r), varname)
                    if not os.path.exists(fp):
                        continue
                    t = ncio.read_tile(fp, varname)
                    for key, out in (("norm", varname + "_normal"), ("se", varname + "_se")):
                        a = t[key]
                        p = ctx.pack_i16(np.where(a == _lib.FILL_F4, 0.0, a.astype(np.float64)))   # :948-957
                        p[a == _lib.FILL_F4] = _lib.FILL_I2
                        ds.variables[out][:, j * ty:(j + 1) * ty, i * tx:(i + 1) * tx] = p
        finally:
            ctx.close()
            ds.close()
        return fpath_out

    def _extent(self, tiles):
        tcols = [int(t[1:3]) for t in tiles]
        trows = [int(t[4:]) for t in tiles]
        ty, tx = self.tinfo.tile_size_y, self.tinfo.tile_size_x
        c0, c1, r0, r1 = min(tcols), max(tcols), min(trows), max(trows)
        return r0, c0, (r1 - r0 + 1) * ty, (c1 - c0 + 1) * tx

    def _place(self, tiles, stores, key, fill, dtype, conv=None):
        r0, c0, ny, nx = self._extent(tiles)
        ty, tx = self.tinfo.tile_size_y, self.tinfo.tile_size_x
        out = None
        for t in tiles:
            if t not in stores:
                continue                                  # "Tile does not exist. Values ... fill values."
            a = stores[t].a[key]
            if conv is not None:
                a = conv(a)
            if out is None:
                out = np.full(a.shape[:-2] + (ny, nx), fill, dtype)
            i, j = (int(t[4:]) - r0) * ty, (int(t[1:3]) - c0) * tx
            out[..., i:i + ty, j:j + tx] = a
        return out

    def create_dly_mosaic(self, tiles, varname, stores):
        """Daily int16 mosaic [ndays, Y, X] (create_dly_ann_mosaics, tiling.py:567-780; the per-year
        split is a slice of the time axis)."""
        return self._place(tiles, stores, "daily_" + varname, _lib.FILL_I2, np.int16)

    def create_normals_mosaic(self, tiles, varname, stores, fpath_out=None, ds_version_str=None, format=None):
        """Monthly normals and kriging standard errors as packed int16 [12, Y, X]
        (tiling.py:948-957: ``np.ma.round(x.astype(float), 2) / SCALE_FACTOR`` cast to int16).  With the reference's
        argument list ``(tiles, varname, path_in, fpath_out, ds_version_str)`` the mosaic is read from / written to
        files (tiling.py:782-971)."""
        if isinstance(stores, (str, os.PathLike)):
            return self._normals_mosaic_file(tiles, varname, os.fspath(stores), fpath_out, ds_version_str, format)
        ctx = _lib.Context(self.device)
        try:
            def conv(a):
                p = ctx.pack_i16(np.where(a == _lib.FILL_F4, 0.0, a.astype(np.float64)))
                p[a == _lib.FILL_F4] = _lib.FILL_I2
                return p
            return (self._place(tiles, stores, "norm_" + varname, _lib.FILL_I2, np.int16, conv),
                    self._place(tiles, stores, "se_" + varname, _lib.FILL_I2, np.int16, conv))
        finally:
            ctx.close()
