"""Serially-complete station database.

Mirrors the reference's ``StationSerialDataDb`` (``twx/db/station_data.py:547-666``): a structured station
table ``stns`` whose fields are the per-station netCDF variables (missing -> NaN, ``station_data.py:159-164``),
the day metadata ``days``, the month row index ``mth_idx`` and the ``(time, station_id)`` float32 observation
matrix ``var``.  A database is opened from its netCDF file -- NetCDF-4 as the reference writes it, or classic
netCDF -- through ``topowx_amd.ncio`` (``StationSerialDataDb(nc_path, var_name, mode=)`` as in the reference), built
from arrays (``topowx_amd.synth``) or loaded from ``.npz``.  The table and the observations live in memory (they are
uploaded to HBM once); a database opened with ``mode='r+'`` keeps its file open as ``ds`` and
``add_stn_variable`` writes through to it (step22:85-112, optimize.py:288-314).
"""
import os

import numpy as np

from .dates import build_mth_idx, get_days_metadata

# Field names: twx/db/station_data.py:49-68
LON = "longitude"
LAT = "latitude"
ELEV = "elevation"
STN_ID = "station_id"
TDI = "tdi"
LST = "lst"
OPTIM_NNGH = "optim_nnghs"
OPTIM_NNGH_ANOM = "optim_nnghs_anom"
MASK = "mask"
VARIO_NUG = "vario_nug"
VARIO_PSILL = "vario_psill"
VARIO_RNG = "vario_rng"
BAD = "bad"
CLIMDIV = "climdiv"
NORM = "norm"


# Name helpers: twx/db/station_data.py:90-124
def get_lst_varname(mth):
    return LST if mth is None else "lst%02d" % mth


def get_norm_varname(mth):
    return NORM if mth is None else "norm%02d" % mth


def get_optim_varname(mth):
    return OPTIM_NNGH if mth is None else "optim_nnghs%02d" % mth


def get_optim_anom_varname(mth):
    return OPTIM_NNGH_ANOM if mth is None else "optim_nnghs_anom%02d" % mth


def get_krigparam_varname(mth, krig_param):
    return krig_param if mth is None else "%s%02d" % (krig_param, mth)


MONTHLY_FIELDS = (
    ("lst", get_lst_varname),
    ("norm", get_norm_varname),
    ("optim_nnghs", get_optim_varname),
    ("optim_nnghs_anom", get_optim_anom_varname),
    ("vario_nug", lambda m: get_krigparam_varname(m, VARIO_NUG)),
    ("vario_psill", lambda m: get_krigparam_varname(m, VARIO_PSILL)),
    ("vario_rng", lambda m: get_krigparam_varname(m, VARIO_RNG)),
)


def stn_dtype(id_len=16):
    dt = [(STN_ID, "U%d" % id_len), (LON, np.float64), (LAT, np.float64),
          (ELEV, np.float64), (TDI, np.float64), (MASK, np.float64),
          (BAD, np.float64), (CLIMDIV, np.float64)]
    for _, namer in MONTHLY_FIELDS:
        dt.extend((namer(m), np.float64) for m in range(1, 13))
    return dt


class StationSerialDataDb(object):
    """Station table + observation matrix for ONE temperature variable.

    Attribute names follow station_data.py:554-616 (``stns``, ``stn_ids``,
    ``days``, ``mth_idx``, ``var``, ``var_name``, ``stn_idxs``).  The station
    table must be sorted by ``station_id``: the reference silently relies on it
    (obs columns come back in DB order, metadata in id order -- SURVEY.md a2).
    """

    def __init__(self, stns, var_name, days=None, obs=None, mode="r"):
        """``stns``: the structured station table -- or, as in the reference (``StationSerialDataDb(nc_path, var_name)``,
        station_data.py:554; step25:53-54, optimize.py:229), the PATH of a station database (NetCDF-4 or classic
        netCDF), read through ``topowx_amd.ncio``.  The file stays open as ``ds`` (read-only, or writable with
        ``mode='r+'`` for ``add_stn_variable``, step22:146-166, step24:88-97); the table and the observations always
        live in memory."""
        self.ds = None
        if isinstance(stns, (str, os.PathLike)):
            from . import ncio
            path = os.fspath(stns)
            stns, _, days, obs = ncio.read_station_db_arrays(path, var_name)
            self.ds = ncio.open_dataset(path, "r" if mode == "r" else "a")   # station_data.py:572 (callers close / sync it: step24:82)
        stns = np.asarray(stns)
        ids = stns[STN_ID]
        if ids.size > 1 and not np.all(ids[1:] > ids[:-1]):
            raise ValueError("station table must be sorted by station_id and unique")
        self.stns = stns
        self.stn_ids = ids.copy()
        self.var_name = var_name
        self.days = days if days is not None else get_days_metadata()
        self.mth_idx = build_mth_idx(self.days)
        if obs is not None:
            obs = np.ascontiguousarray(obs, dtype=np.float32)
            if obs.shape != (self.days.size, stns.size):
                raise ValueError("obs must be [ndays, nstns]")
        self.var = obs
        self.stn_idxs = {sid: i for i, sid in enumerate(self.stn_ids)}
        self.last_stnids = np.array([])
        self.last_obs = None

    def load_obs(self, stn_ids, mth=None):
        """station_data.py:619-666 (columns come back in DB order)."""
        if self.var is None:
            raise ValueError("database holds no observations")
        if isinstance(stn_ids, np.ndarray):
            num_stns = stn_ids.size
            mask = np.nonzero(np.isin(self.stn_ids, stn_ids, assume_unique=True))[0]
            obs = self.var[:, mask]
        else:
            num_stns = 1
            obs = self.var[:, self.stn_idxs[stn_ids]]
        if mth is not None:
            obs = np.take(obs, self.mth_idx[mth], axis=0)
        if num_stns == 1:
            obs = obs.reshape(obs.shape[0])
        return obs

    def add_stn_variable(self, varname, long_name, units, dtype, fill_value=None, reset=True):
        """``add_stn_variable`` (station_data.py:295-341): a new (or reset) per-station variable.  Returns an object
        with the netCDF variable's assignment syntax (``v[i] = x``, ``v[mask] = x``, ``v[:]``): values land in the
        in-memory table ``stns[varname]`` (fill value <-> NaN, as ``_build_stn_struct`` reads them) and, when the
        database was opened from a file with ``mode='r+'``, in the file as well (``ds.sync()`` flushes)."""
        from . import ncio
        dt = np.dtype(dtype)
        key = dt.str[1:]
        fill = ncio.DEFAULT_FILLS.get(key) if fill_value is None else fill_value
        if varname not in self.stns.dtype.names:
            new = np.empty(self.stns.size, dtype=self.stns.dtype.descr + [(varname, np.float64)])
            for f in self.stns.dtype.names:
                new[f] = self.stns[f]
            new[varname] = np.nan
            self.stns = new
        elif reset:
            self.stns[varname] = np.nan
        fvar = None
        if self.ds is not None and getattr(self.ds, "mode", "r") != "r":
            if varname not in self.ds.variables:
                fvar = self.ds.createVariable(varname, dt, (STN_ID,), fill_value=fill)
                fvar.long_name, fvar.units = long_name, units
                if self.ds.data_model != "NETCDF4" and self.stns.size:
                    fvar[:] = fill
            else:
                fvar = self.ds.variables[varname]
                if reset and self.stns.size:
                    fvar[:] = fill
            self.ds.sync()
        return _StnVariable(self, varname, fvar, fill)

    def close(self):
        if self.ds is not None:
            self.ds.close()
            self.ds = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- persistence (npz; the netCDF layout lives in topowx_amd/ncio.py) -----
    def save(self, path, compress=True):
        (np.savez_compressed if compress else np.savez)(path, stns=self.stns, var_name=self.var_name,
                            ymd0=int(self.days.YMD[0]), ymd1=int(self.days.YMD[-1]),
                            obs=self.var if self.var is not None else np.zeros((0, 0), np.float32))

    @classmethod
    def load(cls, path):
        import datetime as dt
        z = np.load(path, allow_pickle=False)

        def _d(ymd):
            return dt.date(ymd // 10000, ymd // 100 % 100, ymd % 100)
        days = get_days_metadata(_d(int(z["ymd0"])), _d(int(z["ymd1"])))
        obs = z["obs"]
        return cls(z["stns"], str(z["var_name"]), days, obs if obs.size else None)


class _StnVariable(object):
    """What ``add_stn_variable`` returns: assignment into the station table and (if open) the file variable."""

    def __init__(self, db, name, fvar, fill):
        self._db, self.name, self._fvar, self._fill = db, name, fvar, fill

    def __setitem__(self, key, value):
        v = np.asarray(np.ma.filled(value, np.nan) if np.ma.isMaskedArray(value) else value, np.float64)
        col = self._db.stns[self.name]
        col[key] = np.where(v == self._fill, np.nan, v) if self._fill is not None else v
        if self._fvar is not None:
            def packed(a):
                a = np.asarray(a, np.float64)
                return (np.where(np.isnan(a), self._fill, a) if self._fill is not None else a).astype(self._fvar.dtype)
            if isinstance(key, (int, np.integer, slice)):
                self._fvar[key] = packed(col[key])              # one station / a run of stations: only that hyperslab (step22:104-110
            else:                                               # assigns station by station: the whole column each time would be O(n^2))
                self._fvar[:] = packed(col)

    def __getitem__(self, key):
        return np.ma.masked_invalid(self._db.stns[self.name][key])

    def __setattr__(self, name, value):
        if name.startswith("_") or name == "name":
            object.__setattr__(self, name, value)
        elif self._fvar is not None:
            self._fvar.setncattr(name, value)


class StationDataWrkChk(StationSerialDataDb):
    """Work-chunk observation cache (interp_tair.py:997-1097).

    The reference preloads the obs of stations inside the chunk's bounding box
    (+3 degrees, grown on a miss) from netCDF.  Here the whole observation
    matrix is already resident (host array / HBM), so ``set_obs`` only records
    the bounds; ``load_obs`` returns the same columns the reference would.
    """

    def __init__(self, stns, var_name, days=None, obs=None, mode="r"):
        StationSerialDataDb.__init__(self, stns, var_name, days, obs, mode)      # (a path works here too: step25:53-54)
        self.chk_bnds = None
        self.chk_deg_buf = None

    def set_obs(self, bnds, deg_buf=3):
        self.chk_bnds = bnds
        self.chk_deg_buf = deg_buf
