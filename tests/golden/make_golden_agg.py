#!/usr/bin/env python3
"""Golden vectors for the monthly / annual aggregation, made by EXECUTING the
reference's own source slices.  Build container only (needs /root/reference):

    python tests/golden/make_golden_agg.py

Slices executed (reference file:lines), read from /root/reference at run time, nothing stored:
  twx/utils/util_dates.py:19-203     date / month metadata helpers
  twx/interp/tiling.py:1080-1166     _TairAggregate (daily_to_mthly, daily_to_ann, mthly_to_ann)
  twx/utils/util_ncdf.py:260-301     GeoNc (geotransform, get_row_col) -> golden_sample_v1.npz (8f-4)

What the slices cannot cover is netCDF4-python (not installed): write_ds_mthly (tiling.py:1169-1219)
reads the daily variable through its auto mask-and-scale (int16 * float32(0.01), _FillValue masked)
and writes the rounded means back through the packing of an 'i2' variable with the same
scale_factor.  Those two steps are restated here in numpy (``unpack`` / ``pack``) exactly as
netCDF4-python documents them; the executed _TairAggregate sits between them.
"""
import builtins
import datetime as dt
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
FILL_I2 = np.int16(-32767)
SCALE = np.float32(0.01)


def _slice(rel, a, b):
    with open(os.path.join(REF, rel)) as fh:
        return "".join(fh.readlines()[a - 1:b])


def load_reference():
    for name, typ in (("bool", bool), ("int", int), ("float", float)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    builtins.long = int                      # py2 name used by util_dates.ymdL
    dates = {}
    exec(compile(_slice("twx/utils/util_dates.py", 19, 203), "util_dates", "exec"), dates)
    til = {"np": np, "YEAR": dates["YEAR"], "MONTH": dates["MONTH"], "get_mth_metadata": dates["get_mth_metadata"]}
    exec(compile(_slice("twx/interp/tiling.py", 1080, 1166), "tiling", "exec"), til)
    return dates, til["_TairAggregate"]


def unpack(raw):
    """netCDF4 auto mask-and-scale of the daily 'i2' variable."""
    return np.ma.masked_array(raw * SCALE, mask=(raw == FILL_I2))


def pack(x):
    """netCDF4 packing of a masked f8 array into the monthly 'i2' variable."""
    x = np.ma.round(x, 2)
    r = np.around(np.ma.getdata(x) / SCALE)
    out = np.where(np.ma.getmaskarray(x), FILL_I2, np.nan_to_num(r)).astype(np.int16)
    return out


def case_inputs(name):
    """Seeded raw int16 daily cubes [ndays, Y, X] and their date range."""
    rng = np.random.default_rng({"two_years": 11, "partial": 12, "one_month": 13}[name])
    if name == "two_years":
        d0, d1, shp = dt.datetime(1999, 1, 1), dt.datetime(2000, 12, 31), (5, 7)
    elif name == "partial":            # starts / ends mid-year: months 3..11 only -> u_mths has 9 entries
        d0, d1, shp = dt.datetime(2003, 3, 15), dt.datetime(2004, 11, 10), (4, 6)
    else:
        d0, d1, shp = dt.datetime(2010, 2, 1), dt.datetime(2010, 2, 28), (3, 5)
    nd = (d1 - d0).days + 1
    doy = np.arange(nd)[:, None, None]
    base = 1200.0 - 1500.0 * np.cos(2 * np.pi * doy / 365.25) + rng.normal(0, 300, (1,) + shp)
    raw = np.rint(base + rng.normal(0, 250, (nd,) + shp)).astype(np.int16)
    # values that are exact in float32 (x.25, x.5, x.75) to exercise rounding ties
    raw[rng.random(raw.shape) < 0.05] = np.int16(25) * rng.integers(-40, 120, 1)[0]
    raw[:, 0, 0] = FILL_I2                               # a masked cell
    raw[rng.random(raw.shape) < 0.01] = FILL_I2          # scattered masked days
    if name == "two_years":
        raw[31:59, 1, 1] = FILL_I2                       # one cell-month entirely masked (Feb 1999)
    return d0, d1, raw


class _FakeDs(object):
    """Stands in for a netCDF4 Dataset holding 1-D lon / lat variables."""

    def __init__(self, lon, lat):
        self.variables = {"lon": lon.copy(), "lat": lat.copy()}


def sample_inputs():
    """Raster axes (north-up) and query points, some exactly on cell edges / centres."""
    rng = np.random.default_rng(21)
    lon = -110.0 + (np.arange(40) + 0.5) * (1.0 / 120.0)
    lat = 45.0 - (np.arange(30) + 0.5) * (1.0 / 120.0)
    qlon = rng.uniform(lon[0] - 0.02, lon[-1] + 0.02, 300)
    qlat = rng.uniform(lat[-1] - 0.02, lat[0] + 0.02, 300)
    qlon[:40], qlat[:40] = lon, lat[:30].repeat(2)[:40]                 # cell centres
    qlon[40:60] = lon[:20] + 0.5 / 120.0                                 # cell edges
    qlat[60:80] = lat[:20] - 0.5 / 120.0
    return lon, lat, qlon, qlat


def make_sample_golden():
    src = {"np": np}
    exec(compile(_slice("twx/utils/util_ncdf.py", 260, 301), "util_ncdf", "exec"), src)
    lon, lat, qlon, qlat = sample_inputs()
    geo = src["GeoNc"](_FakeDs(lon, lat))
    rows, cols, glon, glat = [], [], [], []
    for x, y in zip(qlon, qlat):
        try:
            row, col, gx, gy = geo.get_row_col(x, y)
        except IndexError:                       # lons[col] / lats[row] outside the raster
            row, col, gx, gy = -1, -1, np.nan, np.nan
        rows.append(row); cols.append(col); glon.append(gx); glat.append(gy)
    np.savez_compressed(os.path.join(HERE, "golden_sample_v1.npz"), lon=lon, lat=lat, qlon=qlon, qlat=qlat,
                        row=np.array(rows, np.int32), col=np.array(cols, np.int32), glon=np.array(glon),
                        glat=np.array(glat))
    print("sample", len(rows), "points;", int(np.isnan(glon).sum()), "outside")


def input_hash(raw):
    return hashlib.sha256(np.ascontiguousarray(raw).tobytes()).hexdigest()


def main():
    dates, TairAggregate = load_reference()
    out = {}
    for name in ("two_years", "partial", "one_month"):
        d0, d1, raw = case_inputs(name)
        days = dates["get_days_metadata"](d0, d1)
        tagg = TairAggregate(days)
        tair = unpack(raw)
        mthly = tagg.daily_to_mthly(tair)
        ann = tagg.daily_to_ann(tair)
        ann2 = tagg.mthly_to_ann(mthly)
        assert np.ma.allequal(ann, ann2)
        out[name + "_hash"] = np.array(input_hash(raw))
        out[name + "_year"] = np.asarray(days["YEAR"], np.int32)
        out[name + "_month"] = np.asarray(days["MONTH"], np.int32)
        out[name + "_mthly"] = np.ma.filled(mthly.astype(np.float64), np.nan)
        out[name + "_ann"] = np.ma.filled(ann.astype(np.float64), np.nan)
        out[name + "_mthly_i16"] = pack(mthly)
        # the same aggregation on unscaled f4 / f8 input (masked -> NaN)
        f8 = np.ma.filled(tair.astype(np.float64), np.nan)
        m8 = tagg.daily_to_mthly(np.ma.masked_invalid(f8))
        out[name + "_mthly_f8"] = np.ma.filled(m8.astype(np.float64), np.nan)
        print(name, raw.shape, "groups", mthly.shape[0], "years", ann.shape[0])
    np.savez_compressed(os.path.join(HERE, "golden_agg_v1.npz"), **out)
    make_sample_golden()


if __name__ == "__main__":
    sys.exit(main())
