"""ctypes binding of libtwxhip.so (include/twx.h) -- the only compute path.

There is NO CPU fallback: if the shared library is missing or no GPU is visible
the loader / ``Context`` raise.  (The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import stationdb as sdb
from .dates import MONTH, YEAR

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtwxhip.so")

TMIN, TMAX = 0, 1
VAR_TMIN_BIT, VAR_TMAX_BIT = 1, 2
MAX_NNGHS = 152
MAX_EXCL = 8              # station indices that can be excluded per point (twx_set_exclusions)
FILL_I2 = np.int16(-32767)
FILL_F4 = np.float32(9.969209968386869e36)
FILL_I4 = np.int32(-2147483647)
FLAG_OBS_ADDR64 = 1
FLAG_NO_HOST_SYNC = 2
FLAG_DAILY_GATHER = 4
FLAG_UK_FAST_ONLY = 8     # diagnostic: never use the fp64 covariance build (include/twx.h)
FLAG_UK_F64_ALL = 16      # every kriging system on the fp64 covariance build (include/twx.h)
FLAG_FIX_FULL = 32        # diagnostic: the fixer recomputes every flagged cell's whole series (include/twx.h)
FLAG_NO_TIE_GUARD = 64    # diagnostic: do not re-krige the cells with a day of |Tmax - Tmin| < 2e-5 degC (include/twx.h)
PRECISION_FAST, PRECISION_EXACT = 0, 1     # twx_set_precision

CELL_STATUS = {0: "ok", 1: "too few stations (IndexError, station_select.py:164)",
               2: "Cannot determine the optimal # of neighbors to use!",
               3: "Cannot determine variogram params!",
               4: "floating point error / singular kriging system",
               5: "No valid tmin/tmax in window", 6: "bandwidth above the supported maximum",
               7: "candidate list of the tile overflows (station cluster denser than the library supports)",
               -1: "masked"}

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_sp = C.POINTER(C.c_int16)
_bp = C.POINTER(C.c_uint8)


class TwxParams(C.Structure):
    _fields_ = [("init_nnghs", C.c_int32), ("fixer_tail", C.c_int32), ("norm_yr0", C.c_int32),
                ("norm_yr1", C.c_int32), ("tile_cells", C.c_int32), ("batch_cells", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


class TwxStationTable(C.Structure):
    _fields_ = [("n", C.c_int64), ("lon", _dp), ("lat", _dp), ("elev", _dp), ("tdi", _dp),
                ("lst", _dp), ("norm", _dp), ("optim_nnghs", _dp), ("optim_nnghs_anom", _dp),
                ("vario_nug", _dp), ("vario_psill", _dp), ("vario_rng", _dp), ("obs", _fp)]


class TwxPt(C.Structure):
    _fields_ = [("lon", C.c_double), ("lat", C.c_double), ("elev", C.c_double), ("tdi", C.c_double),
                ("lst", C.c_double * 12)]


PT_DTYPE = np.dtype([("lon", "f8"), ("lat", "f8"), ("elev", "f8"), ("tdi", "f8"), ("lst", "f8", (12,))])
assert PT_DTYPE.itemsize == C.sizeof(TwxPt)


class TwxGrid(C.Structure):
    _fields_ = [("Y", C.c_int32), ("X", C.c_int32), ("mask", C.c_void_p), ("lat", C.c_void_p),
                ("lon", C.c_void_p), ("elev", C.c_void_p), ("tdi", C.c_void_p), ("climdiv", C.c_void_p),
                ("lst_night", C.c_void_p), ("lst_day", C.c_void_p)]


class TwxGridOut(C.Structure):
    _fields_ = [("norm_tmin", C.c_void_p), ("se_tmin", C.c_void_p), ("norm_tmax", C.c_void_p),
                ("se_tmax", C.c_void_p), ("daily_tmin", C.c_void_p), ("daily_tmax", C.c_void_p),
                ("ninvalid", C.c_void_p), ("status", C.c_void_p)]


class TwxRaster(C.Structure):
    _fields_ = [("nrows", C.c_int32), ("ncols", C.c_int32), ("lon", C.POINTER(C.c_double)),
                ("lat", C.POINTER(C.c_double)), ("data", C.POINTER(C.c_float))]


class TwxTiming(C.Structure):
    _fields_ = [("tile_cand_ms", C.c_float), ("select_ms", C.c_float), ("uk_ms", C.c_float),
                ("gwr_ms", C.c_float), ("daily_ms", C.c_float), ("fix_ms", C.c_float),
                ("total_ms", C.c_float), ("cells", C.c_int64), ("uk_solves", C.c_int64),
                ("uk_launches", C.c_int64), ("uk_f64_solves", C.c_int64), ("tie_cells", C.c_int64),
                ("tie_solves", C.c_int64), ("tie_ms", C.c_float), ("deflate_ms", C.c_float)]


class TwxDeflated(C.Structure):
    """twx_deflated (include/twx.h): the chunk streams of a tile whose daily values were deflated on the device."""
    _fields_ = [("data", C.c_void_p * 2), ("offset", C.POINTER(C.c_int64) * 2), ("nchunks", C.c_int32), ("chunk_y", C.c_int32),
                ("chunk_x", C.c_int32), ("reserved", C.c_int32)]


EXPORTS = ("twx_create", "twx_destroy", "twx_last_error", "twx_version", "twx_set_days", "twx_set_stations",
           "twx_knn", "twx_krig_points", "twx_gwr_points", "twx_interp_points", "twx_fix_pair", "twx_pack_i16",
           "twx_interp_grid", "twx_interp_grid_dev", "twx_get_timing", "twx_last_bandwidths",
           "twx_fit_vario_points", "twx_krigall_points", "twx_aggregate_dims", "twx_aggregate", "twx_sample_points", "twx_gwr_xval_points",
           "twx_stream_create", "twx_stream_submit", "twx_stream_wait", "twx_stream_destroy", "twx_stream_times",
           "twx_stream_deflate", "twx_stream_wait_deflated",
           "twx_set_precision", "twx_set_exclusions", "twx_device_memory")

_LIB = None


class TwxError(RuntimeError):
    pass


def load():
    """Load libtwxhip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TwxError("%s not found: build it with ./build.sh (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.twx_last_error.restype = C.c_char_p
        L.twx_version.restype = C.c_char_p
        L.twx_last_error.argtypes = [C.c_void_p]
        L.twx_destroy.argtypes = [C.c_void_p]
        L.twx_destroy.restype = None
        L.twx_last_bandwidths.restype = C.c_int64
        L.twx_stream_destroy.argtypes = [C.c_void_p]
        L.twx_stream_destroy.restype = None
        _LIB = L
    return _LIB


def _p(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


def station_columns(stn_da):
    """Good stations (isnan(bad), interp_tair.py:483-487) as contiguous fp64 SoA."""
    stns = stn_da.stns
    good = np.isnan(stns[sdb.BAD])
    stns = stns[good]
    ids = stns[sdb.STN_ID]
    if ids.size > 1 and not np.all(ids[1:] > ids[:-1]):
        raise ValueError("station table must be sorted by station_id")
    cols = {k: np.ascontiguousarray(stns[n], np.float64) for k, n in
            (("lon", sdb.LON), ("lat", sdb.LAT), ("elev", sdb.ELEV), ("tdi", sdb.TDI))}
    for key, namer in sdb.MONTHLY_FIELDS:
        cols[key] = np.ascontiguousarray(np.stack([stns[namer(m)] for m in range(1, 13)]), np.float64)
    return good, ids, cols


class Context(object):
    """One GPU context (twx_create / twx_destroy)."""

    def __init__(self, device=0, init_nnghs=100, fixer_tail=15, norm_years=(1981, 2010), tile_cells=0,
                 batch_cells=0, flags=0):
        self.lib = load()
        prm = TwxParams(init_nnghs, fixer_tail, norm_years[0], norm_years[1], tile_cells, batch_cells, flags, 0)
        h = C.c_void_p()
        rc = self.lib.twx_create(C.c_int(device), C.byref(prm), C.byref(h))
        if rc != 0:
            raise TwxError("twx_create failed (%d): no usable MI355X / HIP device %d; there is no CPU fallback"
                           % (rc, device))
        self.h = h
        self.device = device
        self.ndays = 0
        self.good = {}
        self.ids = {}
        self.nstn = {}
        self.id_to_idx = {}
        self.mth_days = None
        self._streams = weakref.WeakSet()       # open TileStreams: closed before the context (twx_destroy frees them too)
        self._kept_streams = {}                 # stream(..., keep=True): by shape, reused until drop_streams() / close()

    def drop_streams(self):
        """Close the streams ``stream(..., keep=True)`` holds (each: two device images and its pinned host slots)."""
        for st in list(self._kept_streams.values()):
            st.close()
        self._kept_streams.clear()

    def close(self):
        if getattr(self, "h", None):
            self._kept_streams.clear()
            for st in list(self._streams):
                st.close()
            self.lib.twx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise TwxError("%s failed: %s" % (what, self.lib.twx_last_error(self.h).decode()))

    def set_precision(self, mode):
        """'fast' (default routing: fp32 pair distances, ill-conditioned systems and tie-guard cells on the fp64 covariance
        build) or 'exact' (every kriging system on the fp64 build) for every later call (twx_set_precision)."""
        m = {"fast": PRECISION_FAST, "exact": PRECISION_EXACT}.get(mode, mode)
        self._chk(self.lib.twx_set_precision(self.h, C.c_int(int(m))), "twx_set_precision")

    # ---- model state ------------------------------------------------------------
    def set_days(self, days):
        dm = np.ascontiguousarray(days[MONTH], np.int32)
        dy = np.ascontiguousarray(days[YEAR], np.int32)
        self._chk(self.lib.twx_set_days(self.h, C.c_int64(dm.size), _p(dm, _ip), _p(dy, _ip)), "twx_set_days")
        self.ndays = dm.size
        self.mth_days = [int((dm == m).sum()) for m in range(1, 13)]

    def set_stations(self, var, stn_da, with_obs=True):
        good, ids, cols = station_columns(stn_da)
        obs = None
        if with_obs and stn_da.var is not None:
            if self.ndays != stn_da.days.size:
                self.set_days(stn_da.days)
            obs = np.ascontiguousarray(np.asarray(stn_da.var)[:, good], np.float32)
        t = TwxStationTable()
        t.n = ids.size
        for k in ("lon", "lat", "elev", "tdi", "lst", "norm", "optim_nnghs", "optim_nnghs_anom",
                  "vario_nug", "vario_psill", "vario_rng"):
            setattr(t, k, _p(cols[k], _dp))
        t.obs = _p(obs, _fp)
        self._chk(self.lib.twx_set_stations(self.h, C.c_int(var), C.byref(t)), "twx_set_stations")
        self.good[var], self.ids[var], self.nstn[var] = good, ids, ids.size
        self.id_to_idx[var] = {s: i for i, s in enumerate(ids)}

    # ---- helpers -----------------------------------------------------------------
    @staticmethod
    def make_pts(lon, lat, elev, tdi, lst):
        lon = np.atleast_1d(np.asarray(lon, np.float64))
        pts = np.zeros(lon.size, PT_DTYPE)
        pts["lon"], pts["lat"], pts["elev"], pts["tdi"] = lon, lat, elev, tdi
        pts["lst"] = np.asarray(lst, np.float64).reshape(lon.size, 12)
        return pts

    @staticmethod
    def _i32(a, n, default=None):
        if a is None:
            return None if default is None else np.full(n, default, np.int32)
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.int32), (n,)))

    def _excl(self, excl, n):
        """The ``excl`` argument of a point entry: None, one station index per point ([n] or a scalar), or a LIST of indices
        per point ([n, m], entries < 0 unused; m <= MAX_EXCL: StationSelect's ``stns_rm`` as an array,
        station_select.py:74-103).  Column 0 travels as the entry's own argument, the others through
        ``twx_set_exclusions`` for the call that follows."""
        if excl is None:
            return None
        a = np.asarray(excl, np.int32)
        if a.ndim < 2:
            return np.ascontiguousarray(np.broadcast_to(a, (n,)))
        if a.shape[0] != n:
            a = np.broadcast_to(a, (n, a.shape[-1]))
        if a.shape[1] > MAX_EXCL:
            raise ValueError("at most %d stations can be excluded per point (got %d)" % (MAX_EXCL, a.shape[1]))
        if a.shape[1] > 1:
            more = np.ascontiguousarray(a[:, 1:])
            self._chk(self.lib.twx_set_exclusions(self.h, C.c_int64(n), C.c_int32(more.shape[1]), _p(more, _ip)),
                      "twx_set_exclusions")
        return np.ascontiguousarray(a[:, 0])

    # ---- per-point entries ---------------------------------------------------------
    def knn(self, var, lon, lat, k, excl=None, rm_zero_dist=False):
        lon = np.ascontiguousarray(np.atleast_1d(lon), np.float64)
        lat = np.ascontiguousarray(np.atleast_1d(lat), np.float64)
        n = lon.size
        idx = np.empty((n, k), np.int32)
        dist = np.empty((n, k))
        wgt = np.empty((n, k))
        st = np.empty(n, np.int32)
        ex = self._excl(excl, n)
        self._chk(self.lib.twx_knn(self.h, C.c_int(var), C.c_int64(n), _p(lon, _dp), _p(lat, _dp), C.c_int32(k),
                                   _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(idx, _ip), _p(dist, _dp),
                                   _p(wgt, _dp), _p(st, _ip)), "twx_knn")
        return idx, dist, wgt, st

    def krig_points(self, var, pts, mth, nnghs=None, vario=None, excl=None, rm_zero_dist=False, want_idx=False):
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        mth = self._i32(mth, n)
        nn = self._i32(nnghs, n)
        ex = self._excl(excl, n)
        vp = None
        if vario is not None:
            vp = np.ascontiguousarray(np.broadcast_to(np.asarray(vario, np.float64), (n, 3)))
        mean = np.full(n, np.nan)
        var_ = np.full(n, np.nan)
        used = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        ngh = np.empty((n, MAX_NNGHS), np.int32) if want_idx else None
        self._chk(self.lib.twx_krig_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(mth, _ip), _p(nn, _ip),
                                           _p(vp, _dp), _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(mean, _dp),
                                           _p(var_, _dp), _p(used, _ip), _p(ngh, _ip), _p(st, _ip)),
                  "twx_krig_points")
        return mean, var_, used, st, ngh

    def krigall_points(self, var, pts, mth, nnghs=None, excl=None, rm_zero_dist=False):
        """KrigTairAll.krigall: fit the neighbourhood's variogram, krige with it -- one call, one selection
        (twx_krigall_points).  Returns (mean, variance, vario[n, 3], nnghs_used, status)."""
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        mth = self._i32(mth, n)
        nn = self._i32(nnghs, n)
        ex = self._excl(excl, n)
        mean = np.full(n, np.nan)
        var_ = np.full(n, np.nan)
        vario = np.full((n, 3), np.nan)
        used = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        self._chk(self.lib.twx_krigall_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(mth, _ip), _p(nn, _ip),
                                              _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(mean, _dp), _p(var_, _dp),
                                              _p(vario, _dp), _p(used, _ip), _p(st, _ip)), "twx_krigall_points")
        return mean, var_, vario, used, st

    def fit_vario_points(self, var, pts, mth, nnghs=None, excl=None, rm_zero_dist=False):
        """get_vario_params on the neighbourhood of every (point, month): (nug, psill, range)."""
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        mth = self._i32(mth, n)
        nn = self._i32(nnghs, n)
        ex = self._excl(excl, n)
        vario = np.full((n, 3), np.nan)
        used = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        self._chk(self.lib.twx_fit_vario_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(mth, _ip), _p(nn, _ip),
                                                _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(vario, _dp), _p(used, _ip),
                                                _p(st, _ip)), "twx_fit_vario_points")
        return vario, used, st

    def gwr_points(self, var, pts, pt_norm, mth, nnghs=None, excl=None, rm_zero_dist=False):
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        mth = self._i32(mth, n)
        nn = self._i32(nnghs, n)
        ex = self._excl(excl, n)
        pn = np.ascontiguousarray(np.broadcast_to(np.asarray(pt_norm, np.float64), (n,)))
        ld = max(self.mth_days)
        out = np.full((n, ld), np.nan)
        used = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        self._chk(self.lib.twx_gwr_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(pn, _dp), _p(mth, _ip),
                                          _p(nn, _ip), _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(out, _dp),
                                          C.c_int64(ld), _p(used, _ip), _p(st, _ip)), "twx_gwr_points")
        return out, used, st

    def gwr_xval_points(self, var, pts, pt_norm, mth, nnghs, excl, obs_idx, rm_zero_dist=True):
        """bias / MAE / r^2 of the GWR series of every (point, month, bandwidth) against the observations of
        station ``obs_idx`` (XvalTairAnom.run_xval's statistics, computed on the device)."""
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        mth, nn, ex, oi = self._i32(mth, n), self._i32(nnghs, n), self._excl(excl, n), self._i32(obs_idx, n)
        pn = np.ascontiguousarray(np.broadcast_to(np.asarray(pt_norm, np.float64), (n,)))
        bias, mae, r2 = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
        used = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        self._chk(self.lib.twx_gwr_xval_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(pn, _dp), _p(mth, _ip),
                                               _p(nn, _ip), _p(ex, _ip), C.c_int(int(rm_zero_dist)), _p(oi, _ip),
                                               _p(bias, _dp), _p(mae, _dp), _p(r2, _dp), _p(used, _ip), _p(st, _ip)),
                  "twx_gwr_xval_points")
        return bias, mae, r2, used, st

    def interp_points(self, var, pts, excl=None, rm_zero_dist=False, daily=True):
        pts = np.ascontiguousarray(pts, PT_DTYPE)
        n = pts.size
        ex = self._excl(excl, n)
        d = np.full((n, self.ndays), np.nan) if daily else None
        norms = np.full((n, 12), np.nan)
        se = np.full((n, 12), np.nan)
        st = np.zeros(n, np.int32)
        self._chk(self.lib.twx_interp_points(self.h, C.c_int(var), C.c_int64(n), _p(pts), _p(ex, _ip),
                                             C.c_int(int(rm_zero_dist)), _p(d, _dp), _p(norms, _dp), _p(se, _dp),
                                             _p(st, _ip)), "twx_interp_points")
        return d, norms, se, st

    def fix_pair(self, tmin, tmax):
        tmin = np.array(np.atleast_2d(tmin), np.float64, order="C")
        tmax = np.array(np.atleast_2d(tmax), np.float64, order="C")
        n = tmin.shape[0]
        if tmin.shape[1] != self.ndays:
            raise ValueError("series length must equal the day axis set with set_days")
        ninv = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        nmin = np.full((n, 12), np.nan)
        nmax = np.full((n, 12), np.nan)
        self._chk(self.lib.twx_fix_pair(self.h, C.c_int64(n), _p(tmin, _dp), _p(tmax, _dp), _p(ninv, _ip),
                                        _p(nmin, _dp), _p(nmax, _dp), _p(st, _ip)), "twx_fix_pair")
        return tmin, tmax, ninv, nmin, nmax, st

    def pack_i16(self, x):
        x = np.ascontiguousarray(x, np.float64)
        out = np.empty(x.shape, np.int16)
        self._chk(self.lib.twx_pack_i16(self.h, C.c_int64(x.size), _p(x, _dp), _p(out, _sp)), "twx_pack_i16")
        return out

    # ---- monthly / annual aggregation ------------------------------------
    AGG_DTYPES = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}

    def aggregate_dims(self):
        nyr, nmth = C.c_int32(), C.c_int32()
        self._chk(self.lib.twx_aggregate_dims(self.h, C.byref(nyr), C.byref(nmth)), "twx_aggregate_dims")
        return nyr.value, nmth.value

    def aggregate(self, daily, mthly=True, mthly_i16=False, ann=False):
        """daily [ndays, ...] (raw int16 / f4 / f8 with NaN = masked) on the day axis of ``set_days`` ->
        dict of the requested f8 ``mthly`` [nyr*nmth, ...], int16 ``mthly_i16`` and f8 ``ann`` [nyr, ...]."""
        daily = np.ascontiguousarray(daily)
        if daily.dtype not in self.AGG_DTYPES:
            raise TypeError("aggregate: daily must be int16, float32 or float64")
        nyr, nmth = self.aggregate_dims()
        shp = daily.shape[1:]
        ncell = int(np.prod(shp, dtype=np.int64))
        out = {}
        if mthly:
            out["mthly"] = np.empty((nyr * nmth,) + shp, np.float64)
        if mthly_i16:
            out["mthly_i16"] = np.empty((nyr * nmth,) + shp, np.int16)
        if ann:
            out["ann"] = np.empty((nyr,) + shp, np.float64)
        ms = C.c_float()
        self._chk(self.lib.twx_aggregate(
            self.h, C.c_void_p(daily.ctypes.data), C.c_int(self.AGG_DTYPES[daily.dtype]), C.c_int64(ncell),
            C.c_int(0), _p(out["mthly"], _dp) if mthly else None, _p(out["mthly_i16"], _sp) if mthly_i16 else None,
            _p(out["ann"], _dp) if ann else None, C.c_void_p(0), C.byref(ms)), "twx_aggregate")
        out["kernel_ms"] = ms.value
        return out

    def aggregate_dev(self, daily_ptr, dtype, ncell, mthly_ptr=0, mthly_i16_ptr=0, ann_ptr=0, stream=0, timed=True):
        """Device-pointer form; returns the kernel time in ms when ``timed`` (synchronises)."""
        ms = C.c_float()
        self._chk(self.lib.twx_aggregate(
            self.h, C.c_void_p(daily_ptr), C.c_int(dtype), C.c_int64(ncell), C.c_int(1),
            C.cast(C.c_void_p(mthly_ptr), _dp), C.cast(C.c_void_p(mthly_i16_ptr), _sp),
            C.cast(C.c_void_p(ann_ptr), _dp), C.c_void_p(stream), C.byref(ms) if timed else None), "twx_aggregate")
        return ms.value if timed else None

    # ---- point-mode predictor sampling --------------------------------------
    def sample_points(self, lons, lats, data, lon, lat, order=0, missing=-9999.0):
        """Raster [nrows, ncols] (north-up, NaN = missing) at the points -> (val, row, col, status)."""
        lons, lats = np.ascontiguousarray(lons, np.float64), np.ascontiguousarray(lats, np.float64)
        data = np.ascontiguousarray(data, np.float32)
        if data.shape != (lats.size, lons.size):
            raise ValueError("sample_points: data must be [lat, lon]")
        lon = np.ascontiguousarray(np.atleast_1d(lon), np.float64)
        lat = np.ascontiguousarray(np.atleast_1d(lat), np.float64)
        n = lon.size
        r = TwxRaster(lats.size, lons.size, _p(lons, _dp), _p(lats, _dp), _p(data, _fp))
        val, row, col, st = np.empty(n), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        self._chk(self.lib.twx_sample_points(self.h, C.byref(r), C.c_int64(n), _p(lon, _dp), _p(lat, _dp),
                                             C.c_int(order), C.c_double(missing), _p(val, _dp), _p(row, _ip),
                                             _p(col, _ip), _p(st, _ip)), "twx_sample_points")
        return val, row, col, st

    # ---- grid entries ------------------------------------------------------------------
    @staticmethod
    def grid_arrays(grid, rows=None, cols=None):
        rs = rows if rows is not None else slice(None)
        cs = cols if cols is not None else slice(None)
        a = dict(lat=np.ascontiguousarray(grid["lat"][rs], np.float64),
                 lon=np.ascontiguousarray(grid["lon"][cs], np.float64),
                 mask=np.ascontiguousarray(grid["mask"][rs, cs], np.uint8),
                 elev=np.ascontiguousarray(grid["elev"][rs, cs], np.float32),
                 tdi=np.ascontiguousarray(grid["tdi"][rs, cs], np.float32),
                 climdiv=np.ascontiguousarray(grid["climdiv"][rs, cs], np.int32),
                 lst_night=np.ascontiguousarray(grid["lst_night"][:, rs, cs], np.float32),
                 lst_day=np.ascontiguousarray(grid["lst_day"][:, rs, cs], np.float32))
        return a

    def interp_grid(self, grid, variables=("tmin", "tmax"), daily=False, rows=None, cols=None):
        """step25 worker loop on host arrays; returns result arrays pre-filled with
        the netCDF fill values the reference uses (step25:68-88)."""
        a = self.grid_arrays(grid, rows, cols)
        Y, X = a["mask"].shape
        g = TwxGrid(Y, X, *[a[k].ctypes.data for k in ("mask", "lat", "lon", "elev", "tdi", "climdiv",
                                                     "lst_night", "lst_day")])
        out = {}
        vars_mask = 0
        for v, bit in (("tmin", VAR_TMIN_BIT), ("tmax", VAR_TMAX_BIT)):
            if v not in variables:
                continue
            vars_mask |= bit
            out["norm_" + v] = np.full((12, Y, X), FILL_F4, np.float32)
            out["se_" + v] = np.full((12, Y, X), FILL_F4, np.float32)
            if daily:
                out["daily_" + v] = np.full((self.ndays, Y, X), FILL_I2, np.int16)
        out["ninvalid"] = np.full((Y, X), FILL_I4, np.int32)
        out["status"] = np.full((Y, X), -1, np.int32)
        o = TwxGridOut(*[out[k].ctypes.data if k in out else None for k in
                         ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax", "daily_tmin", "daily_tmax",
                          "ninvalid", "status")])
        self._chk(self.lib.twx_interp_grid(self.h, C.byref(g), C.byref(o), C.c_int(vars_mask)), "twx_interp_grid")
        return out

    def stream(self, Y, X, variables=("tmin", "tmax"), daily=False, nslots=2, deflate_chunks=None, keep=False):
        """Streamed tiles of one shape (twx_stream_*): see ``TileStream``.  ``keep``: the stream stays with the context and the
        next call with the same arguments gets it back (pinning and unpinning the host slots of a configs[3]-sized stream takes
        ~1 s each way: a run that streams tile lists one after the other pays it once)."""
        if not keep:
            return TileStream(self, Y, X, variables, daily, nslots, deflate_chunks)
        key = (Y, X, tuple(variables), bool(daily), int(nslots), None if deflate_chunks is None else tuple(int(v) for v in deflate_chunks))
        st = self._kept_streams.get(key)
        if st is None or not getattr(st, "h", None):
            st = self._kept_streams[key] = TileStream(self, Y, X, variables, daily, nslots, deflate_chunks)
        return st

    def interp_grid_dev(self, g, o, vars_mask, stream=0):
        """Device-pointer entry (TwxGrid / TwxGridOut hold device addresses)."""
        self._chk(self.lib.twx_interp_grid_dev(self.h, C.byref(g), C.byref(o), C.c_int(vars_mask),
                                               C.c_void_p(stream)), "twx_interp_grid_dev")

    def last_bandwidths(self, var, max_cells=1 << 22):
        """nnghs[cells, 12] of the last device batch (diagnostic)."""
        buf = np.zeros((max_cells, 12), np.int32)
        n = self.lib.twx_last_bandwidths(self.h, C.c_int(var), _p(buf, _ip), C.c_int64(buf.size))
        if n < 0:
            self._chk(-1, "twx_last_bandwidths")
        return buf[:min(n, max_cells)]

    def device_memory(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_int64(), C.c_int64()
        self._chk(self.lib.twx_device_memory(self.h, C.byref(f), C.byref(t)), "twx_device_memory")
        return f.value, t.value

    def timing(self):
        t = TwxTiming()
        self._chk(self.lib.twx_get_timing(self.h, C.byref(t)), "twx_get_timing")
        return {k: getattr(t, k) for k, _ in TwxTiming._fields_}


class TileStream(object):
    """Pipelined tiles (twx_stream_create / submit / wait): the outputs of tile t travel to pinned host memory while
    the kernels of tile t + 1 run.  ``wait(slot)`` returns numpy views of the slot's pinned block; they stay valid
    until the slot is submitted again (copy or write them out before that).

    ``deflate_chunks=(cy, cx)`` (twx_stream_deflate): the daily values leave the GPU as the chunk bytes of an HDF5 dataset with
    shuffle + deflate (one zlib stream per variable and ``(ndays, cy, cx)`` chunk, formed on the device); ``wait`` then returns
    ``deflated_tmin`` / ``deflated_tmax`` -- lists of uint8 views, chunks in row-major order -- instead of the daily arrays,
    and ``deflate_chunks``."""

    def __init__(self, ctx, Y, X, variables=("tmin", "tmax"), daily=False, nslots=2, deflate_chunks=None):
        self.ctx, self.Y, self.X, self.daily, self.nslots = ctx, Y, X, daily, nslots
        self.deflate_chunks = None
        self.vars_mask = (VAR_TMIN_BIT if "tmin" in variables else 0) | (VAR_TMAX_BIT if "tmax" in variables else 0)
        h = C.c_void_p()
        ctx._chk(ctx.lib.twx_stream_create(ctx.h, C.c_int(Y), C.c_int(X), C.c_int(self.vars_mask), C.c_int(int(daily)),
                                           C.c_int(nslots), C.byref(h)), "twx_stream_create")
        self.h = h
        ctx._streams.add(self)
        if deflate_chunks is not None:
            cy, cx = (int(v) for v in deflate_chunks)
            ctx._chk(ctx.lib.twx_stream_deflate(h, C.c_int(cy), C.c_int(cx)), "twx_stream_deflate")
            self.deflate_chunks = (cy, cx)

    def submit(self, slot, grid, rows=None, cols=None):
        a = Context.grid_arrays(grid, rows, cols)
        if a["mask"].shape != (self.Y, self.X):
            raise ValueError("tile shape differs from the stream's")
        g = TwxGrid(self.Y, self.X, *[a[k].ctypes.data for k in ("mask", "lat", "lon", "elev", "tdi", "climdiv",
                                                                 "lst_night", "lst_day")])
        self.ctx._chk(self.ctx.lib.twx_stream_submit(self.h, C.c_int(slot), C.byref(g)), "twx_stream_submit")

    def wait(self, slot):
        o = TwxGridOut()
        ms = C.c_float()
        df = TwxDeflated()
        if self.deflate_chunks:
            self.ctx._chk(self.ctx.lib.twx_stream_wait_deflated(self.h, C.c_int(slot), C.byref(o), C.byref(df), C.byref(ms)),
                          "twx_stream_wait_deflated")
        else:
            self.ctx._chk(self.ctx.lib.twx_stream_wait(self.h, C.c_int(slot), C.byref(o), C.byref(ms)), "twx_stream_wait")
        Y, X, nd = self.Y, self.X, self.ctx.ndays
        spec = (("norm_tmin", np.float32, (12, Y, X)), ("se_tmin", np.float32, (12, Y, X)),
                ("norm_tmax", np.float32, (12, Y, X)), ("se_tmax", np.float32, (12, Y, X)),
                ("daily_tmin", np.int16, (nd, Y, X)), ("daily_tmax", np.int16, (nd, Y, X)),
                ("ninvalid", np.int32, (Y, X)), ("status", np.int32, (Y, X)))
        out = {}
        for name, dt, shape in spec:
            ptr = getattr(o, name)
            if ptr:
                n = int(np.prod(shape, dtype=np.int64))
                buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
                out[name] = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
        out["device_ms"] = ms.value
        if self.deflate_chunks:
            out["deflate_chunks"] = self.deflate_chunks
            for v, name in enumerate(("tmin", "tmax")):
                if df.data[v]:
                    off = np.ctypeslib.as_array(df.offset[v], shape=(df.nchunks + 1,))
                    whole = np.frombuffer((C.c_char * int(off[-1])).from_address(df.data[v]), np.uint8)
                    out["deflated_" + name] = [whole[off[c]:off[c + 1]] for c in range(df.nchunks)]
        return out

    def times(self, slot):
        """(device_ms, copy_ms) of the tile last waited for in ``slot``: its kernels, and its copy-out to pinned memory."""
        dev, cp = C.c_float(), C.c_float()
        self.ctx._chk(self.ctx.lib.twx_stream_times(self.h, C.c_int(slot), C.byref(dev), C.byref(cp)), "twx_stream_times")
        return dev.value, cp.value

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):            # (a closed context has destroyed its streams: the handle is dead)
                self.ctx.lib.twx_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
