"""Mosaicking and monthly / annual aggregation of the tile product.

Array-level counterparts of ``twx/interp/tiling.py``: ``TileMosaic`` (:553-971), ``_TairAggregate``
(:1080-1166) and ``write_ds_mthly`` (:1169-1219, scripts/step27_create_monthly.py).  The netCDF
containers are SURVEY.md 8f-2; the arithmetic between them -- the means over (year, month) groups, the
rounding and the int16 packing -- runs in libtwxhip (``twx_aggregate``, ``twx_pack_i16``).
"""
import numpy as np

from .. import _lib
from ..dates import MONTH, YEAR, get_mth_metadata

__all__ = ["TairAggregate", "TileMosaic", "mthly_from_daily"]


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


class TileMosaic(object):
    """Assemble tile results (``step25.TileStore`` objects keyed by tile id) into mosaics
    (tiling.py:553-971).  The mosaic spans the tile rows / columns between the first and the last tile
    given (tiling.py:570-596); tiles that are missing stay at the fill value (:772-776)."""

    def __init__(self, tile_grid_info, device=0):
        self.tinfo = tile_grid_info
        self.device = device

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

    def create_normals_mosaic(self, tiles, varname, stores):
        """Monthly normals and kriging standard errors as packed int16 [12, Y, X]
        (tiling.py:948-957: ``np.ma.round(x.astype(float), 2) / SCALE_FACTOR`` cast to int16)."""
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
