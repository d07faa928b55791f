"""In-memory serially-complete station database.

Mirrors the *schema* of the reference's ``StationSerialDataDb``
(``twx/db/station_data.py:547-666``): a structured station table ``stns`` whose
fields are the per-station netCDF variables (missing -> NaN,
``station_data.py:159-164``), the day metadata ``days``, the month row index
``mth_idx`` and the ``(time, station_id)`` float32 observation matrix ``var``.
netCDF I/O is out of scope for the hot path; a database
is built from arrays (``topowx_amd.synth``) or loaded from ``.npz``.
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
        station_data.py:554; step25:53-54, optimize.py:229), the PATH of a station database, read through
        ``topowx_amd.ncio`` (classic / 64-bit-offset netCDF; ``mode`` is accepted for call-site parity, the table
        lives in memory and is written back with ``ncio.write_station_db``)."""
        if isinstance(stns, (str, os.PathLike)):
            from . import ncio
            stns, _, days, obs = ncio.read_station_db_arrays(os.fspath(stns), var_name)
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
