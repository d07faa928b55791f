"""Day / month metadata for the interpolation period.

Python-3 counterpart of the few date helpers the hot path needs from the
reference (``twx/utils/util_dates.py:150-169``: ``get_days_metadata_dates`` and
``get_mth_metadata``).  Only the integer YEAR / MONTH / DAY / YMD columns are
kept; the reference's ``DATE`` object column is never read on the path.
"""
import datetime as _dt

import numpy as np

DATE = "DATE"
YEAR = "YEAR"
MONTH = "MONTH"
DAY = "DAY"
YDAY = "YDAY"
YMD = "YMD"

_DAYS_DTYPE = [(YEAR, np.int32), (MONTH, np.int32), (DAY, np.int32),
               (YDAY, np.int32), (YMD, np.int32)]
_MTHS_DTYPE = [(YEAR, np.int32), (MONTH, np.int32), (YMD, np.int32)]


def get_days_metadata(start=_dt.date(1948, 1, 1), end=_dt.date(2016, 12, 31)):
    """Record array with one row per day in [start, end] (inclusive)."""
    if isinstance(start, _dt.datetime):
        start = start.date()
    if isinstance(end, _dt.datetime):
        end = end.date()
    d64 = np.arange(np.datetime64(start), np.datetime64(end) + 1,
                    dtype="datetime64[D]")
    yrs = d64.astype("datetime64[Y]")
    mths = d64.astype("datetime64[M]")
    days = np.recarray(d64.size, dtype=_DAYS_DTYPE)
    days[YEAR] = yrs.astype(np.int64) + 1970
    days[MONTH] = mths.astype(np.int64) % 12 + 1
    days[DAY] = (d64 - mths.astype("datetime64[D]")).astype(np.int64) + 1
    days[YDAY] = (d64 - yrs.astype("datetime64[D]")).astype(np.int64) + 1
    days[YMD] = days[YEAR] * 10000 + days[MONTH] * 100 + days[DAY]
    return days


def get_mth_metadata(str_yr, end_yr):
    """One row per (year, month) for str_yr..end_yr (util_dates.py:161-169)."""
    yrs = np.repeat(np.arange(str_yr, end_yr + 1), 12)
    mths = np.tile(np.arange(1, 13), end_yr - str_yr + 1)
    out = np.recarray(yrs.size, dtype=_MTHS_DTYPE)
    out[YEAR] = yrs
    out[MONTH] = mths
    out[YMD] = yrs * 10000 + mths * 100 + 1
    return out


def build_mth_idx(days):
    """``mth_idx`` dict of the reference DB object (station_data.py:572-576)."""
    idx = {m: np.nonzero(days[MONTH] == m)[0] for m in range(1, 13)}
    idx[None] = np.arange(days.size)
    return idx
