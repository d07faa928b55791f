"""ctypes binding of the CPU oracle (oracle/libtwxoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MONTHLY = ("lst", "norm", "optim_nnghs", "optim_nnghs_anom", "vario_nug",
           "vario_psill", "vario_rng")

ERR_NAMES = {0: "ok", 1: "few_stations", 2: "nnghs", 3: "vario", 4: "numeric",
             5: "fixer", 6: "range"}

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class OrcDb(C.Structure):
    _fields_ = [("n", C.c_int64), ("lon", _dp), ("lat", _dp), ("elev", _dp), ("tdi", _dp),
                ("lst", _dp), ("norm", _dp), ("optim_nnghs", _dp), ("optim_nnghs_anom", _dp),
                ("vario_nug", _dp), ("vario_psill", _dp), ("vario_rng", _dp),
                ("obs", _fp), ("ndays", C.c_int64), ("day_month", _ip), ("day_year", _ip)]


class OrcPt(C.Structure):
    _fields_ = [("lon", C.c_double), ("lat", C.c_double), ("elev", C.c_double),
                ("tdi", C.c_double), ("lst", C.c_double * 12)]


class OrcParams(C.Structure):
    _fields_ = [("init_nnghs", C.c_int32), ("fixer_tail", C.c_int32),
                ("norm_yr0", C.c_int32), ("norm_yr1", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libtwxoracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtwxoracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_grt_circle_dist.restype = C.c_double
        L.orc_grt_circle_dist.argtypes = [C.c_double] * 4
        L.orc_ellip_dist.restype = C.c_double
        L.orc_ellip_dist.argtypes = [C.c_double] * 4
        L.orc_nearest.restype = C.c_int64
        _LIB = L
    return _LIB


def _ptr(a, t):
    return a.ctypes.data_as(t)


def soa_from_stns(stns):
    """SoA fp64 columns of a structured station table (topowx_amd.stationdb)."""
    from topowx_amd import stationdb as sdb
    out = {k: np.ascontiguousarray(stns[n], np.float64) for k, n in
           (("lon", sdb.LON), ("lat", sdb.LAT), ("elev", sdb.ELEV), ("tdi", sdb.TDI))}
    for key, namer in sdb.MONTHLY_FIELDS:
        out[key] = np.ascontiguousarray(np.stack([stns[namer(m)] for m in range(1, 13)]), np.float64)
    return out


class Db(object):
    """Holds the arrays an ``orc_db`` points into."""

    def __init__(self, stn_da, good_only=True):
        from topowx_amd import stationdb as sdb
        from topowx_amd.dates import MONTH, YEAR
        stns = stn_da.stns
        self.good = np.isnan(stns[sdb.BAD]) if good_only else np.ones(stns.size, bool)
        stns = stns[self.good]
        self.cols = soa_from_stns(stns)
        self.n = stns.size
        self.obs = None
        if stn_da.var is not None:
            self.obs = np.ascontiguousarray(stn_da.var[:, self.good], np.float32)
        self.day_month = np.ascontiguousarray(stn_da.days[MONTH], np.int32)
        self.day_year = np.ascontiguousarray(stn_da.days[YEAR], np.int32)
        self.ndays = self.day_month.size
        s = OrcDb()
        s.n = self.n
        for k in ("lon", "lat", "elev", "tdi") + MONTHLY:
            setattr(s, k, _ptr(self.cols[k], _dp))
        s.obs = _ptr(self.obs, _fp) if self.obs is not None else None
        s.ndays = self.ndays
        s.day_month = _ptr(self.day_month, _ip)
        s.day_year = _ptr(self.day_year, _ip)
        self.c = s


def params(init_nnghs=100, fixer_tail=15, yr0=1981, yr1=2010):
    return OrcParams(init_nnghs, fixer_tail, yr0, yr1)


def make_pt(lon, lat, elev, tdi, lst12):
    p = OrcPt()
    p.lon, p.lat, p.elev, p.tdi = float(lon), float(lat), float(elev), float(tdi)
    for m in range(12):
        p.lst[m] = float(lst12[m])
    return p


def grt_circle_dist(lon1, lat1, lon2, lat2):
    L = lib()
    f = np.vectorize(lambda a, b, c, d: L.orc_grt_circle_dist(a, b, c, d), otypes=[np.float64])
    return f(lon1, lat1, lon2, lat2)


def ellip_dist(lon1, lat1, lon2, lat2):
    L = lib()
    f = np.vectorize(lambda a, b, c, d: L.orc_ellip_dist(a, b, c, d), otypes=[np.float64])
    return f(lon1, lat1, lon2, lat2)


class exclusions(object):
    """``with exclusions(indices): ...`` -- ``stns_rm`` as an array of ids (station_select.py:74-103): every oracle call of this
    thread inside the block also drops these station indices (entries < 0 are ignored)."""

    def __init__(self, idx):
        self.idx = np.ascontiguousarray([i for i in np.asarray(idx).ravel() if i >= 0], np.int32)

    def __enter__(self):
        lib().orc_set_exclusions(C.c_int(self.idx.size), _ptr(self.idx, _ip))
        return self

    def __exit__(self, *exc):
        lib().orc_set_exclusions(C.c_int(0), None)


def nearest(db, lat, lon, ksel, excl=-1, rm_zero_dist=False):
    idx = np.zeros(ksel, np.int32)
    dist = np.zeros(ksel, np.float64)
    n = lib().orc_nearest(C.byref(db.c), C.c_double(lat), C.c_double(lon), C.c_int32(excl),
                          C.c_int(int(rm_zero_dist)), C.c_int64(ksel), _ptr(idx, _ip), _ptr(dist, _dp))
    return idx[:n], dist[:n]


def select(db, lat, lon, k, excl=-1, rm_zero_dist=False):
    """(status, idx, dist, wgt) of StationSelect.set_ngh_stns."""
    nidx, ndist = nearest(db, lat, lon, k + 1, excl, rm_zero_dist)
    idx = np.zeros(k, np.int32)
    dist = np.zeros(k)
    wgt = np.zeros(k)
    rc = lib().orc_select(_ptr(nidx, _ip), _ptr(ndist, _dp), C.c_int64(nidx.size), C.c_int(k),
                          _ptr(idx, _ip), _ptr(dist, _dp), _ptr(wgt, _dp))
    return rc, idx, dist, wgt


def uk(lon, lat, elev, lst, y, pt, nug, psill, rng):
    lon, lat, elev, lst, y = (np.ascontiguousarray(a, np.float64) for a in (lon, lat, elev, lst, y))
    mean, var = C.c_double(), C.c_double()
    rc = lib().orc_uk(C.c_int(lon.size), _ptr(lon, _dp), _ptr(lat, _dp), _ptr(elev, _dp),
                      _ptr(lst, _dp), _ptr(y, _dp), C.c_double(pt[0]), C.c_double(pt[1]),
                      C.c_double(pt[2]), C.c_double(pt[3]), C.c_double(nug), C.c_double(psill),
                      C.c_double(rng), C.byref(mean), C.byref(var))
    return rc, mean.value, var.value


def gwr_hat(X5, w, x5):
    X5 = np.ascontiguousarray(X5, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    x5 = np.ascontiguousarray(x5, np.float64)
    z = np.zeros(w.size)
    rc = lib().orc_gwr_hat(C.c_int(w.size), _ptr(X5, _dp), _ptr(w, _dp), _ptr(x5, _dp), _ptr(z, _dp))
    return rc, z


def krig(db, prm, pt, mth, nnghs=0, vario=None, excl=-1, rm_zero_dist=False):
    mean, var, used = C.c_double(), C.c_double(), C.c_int()
    idx = np.zeros(512, np.int32)
    v = None
    if vario is not None:
        v = (C.c_double * 3)(*vario)
    rc = lib().orc_krig(C.byref(db.c), C.byref(prm), C.byref(pt), C.c_int(mth), C.c_int(nnghs), v,
                        C.c_int32(excl), C.c_int(int(rm_zero_dist)), C.byref(mean), C.byref(var),
                        C.byref(used), _ptr(idx, _ip))
    return rc, mean.value, var.value, used.value, idx[:max(used.value, 0)]


def gwr_mth(db, prm, pt, pt_norm, mth, nnghs=0, excl=-1, rm_zero_dist=False):
    nd = int((db.day_month == mth).sum())
    out = np.zeros(nd)
    z = np.zeros(512)
    idx = np.zeros(512, np.int32)
    used = C.c_int()
    rc = lib().orc_gwr_mth(C.byref(db.c), C.byref(prm), C.byref(pt), C.c_double(pt_norm), C.c_int(mth),
                           C.c_int(nnghs), C.c_int32(excl), C.c_int(int(rm_zero_dist)), _ptr(out, _dp),
                           C.byref(used), _ptr(z, _dp), _ptr(idx, _ip))
    k = max(used.value, 0)
    return rc, out, k, z[:k], idx[:k]


def interp(db, prm, pt, excl=-1, rm_zero_dist=False, daily=True):
    d = np.zeros(db.ndays) if daily else None
    norms = np.zeros(12)
    se = np.zeros(12)
    rc = lib().orc_interp(C.byref(db.c), C.byref(prm), C.byref(pt), C.c_int32(excl),
                          C.c_int(int(rm_zero_dist)), _ptr(d, _dp) if daily else None,
                          _ptr(norms, _dp), _ptr(se, _dp))
    return rc, d, norms, se


def fixer(tmin, tmax, tail=15):
    tmin = np.array(tmin, np.float64)
    tmax = np.array(tmax, np.float64)
    ninv = C.c_int32()
    rc = lib().orc_fixer(_ptr(tmin, _dp), _ptr(tmax, _dp), C.c_int64(tmin.size), C.c_int(tail), C.byref(ninv))
    return rc, tmin, tmax, ninv.value


def recompute_norms(daily, day_month, day_year, yr0=1981, yr1=2010):
    daily = np.ascontiguousarray(daily, np.float64)
    dm = np.ascontiguousarray(day_month, np.int32)
    dy = np.ascontiguousarray(day_year, np.int32)
    out = np.zeros(12)
    lib().orc_recompute_norms(_ptr(daily, _dp), C.c_int64(daily.size), _ptr(dm, _ip), _ptr(dy, _ip),
                              C.c_int(yr0), C.c_int(yr1), _ptr(out, _dp))
    return out


def pack_i16(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(x.size, np.int16)
    lib().orc_pack_i16(_ptr(x, _dp), C.c_int64(x.size), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out.reshape(x.shape)


FILL_I2 = np.int16(-32767)
FILL_F4 = np.float32(9.969209968386869e36)
FILL_I4 = np.int32(-2147483647)


def interp_grid(db_tmin, db_tmax, prm, grid, daily=False, nthreads=1, rows=None, cols=None):
    """step25 worker loop over a grid (or the [rows, cols] sub-window)."""
    rs = rows if rows is not None else slice(0, grid["lat"].size)
    cs = cols if cols is not None else slice(0, grid["lon"].size)
    lat = np.ascontiguousarray(grid["lat"][rs], np.float64)
    lon = np.ascontiguousarray(grid["lon"][cs], np.float64)
    Y, X = lat.size, lon.size
    mask = np.ascontiguousarray(grid["mask"][rs, cs], np.uint8)
    elev = np.ascontiguousarray(grid["elev"][rs, cs], np.float32)
    tdi = np.ascontiguousarray(grid["tdi"][rs, cs], np.float32)
    lst_n = np.ascontiguousarray(grid["lst_night"][:, rs, cs], np.float32)
    lst_d = np.ascontiguousarray(grid["lst_day"][:, rs, cs], np.float32)
    out = {}
    for v, db in (("tmin", db_tmin), ("tmax", db_tmax)):
        if db is None:
            continue
        out["norm_" + v] = np.full((12, Y, X), FILL_F4, np.float32)
        out["se_" + v] = np.full((12, Y, X), FILL_F4, np.float32)
        if daily:
            out["daily_" + v] = np.full((db.ndays, Y, X), FILL_I2, np.int16)
    out["ninvalid"] = np.full((Y, X), FILL_I4, np.int32)
    out["status"] = np.full((Y, X), -1, np.int32)
    i16p = C.POINTER(C.c_int16)

    def g(name, t):
        return _ptr(out[name], t) if name in out else None
    lib().orc_interp_grid(
        C.byref(db_tmin.c) if db_tmin is not None else None,
        C.byref(db_tmax.c) if db_tmax is not None else None,
        C.byref(prm), C.c_int(Y), C.c_int(X), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
        _ptr(lat, _dp), _ptr(lon, _dp), _ptr(elev, _fp), _ptr(tdi, _fp), _ptr(lst_n, _fp), _ptr(lst_d, _fp),
        g("norm_tmin", _fp), g("se_tmin", _fp), g("norm_tmax", _fp), g("se_tmax", _fp),
        g("daily_tmin", i16p), g("daily_tmax", i16p), _ptr(out["ninvalid"], _ip), _ptr(out["status"], _ip),
        C.c_int(nthreads))
    return out


# ---- second tier (8f-1): variogram estimation + range fit --------------------------------------
def get_vario_params(lon, lat, elev, lst, y, max_ngh_dist):
    lon, lat, elev, lst, y = (np.ascontiguousarray(a, np.float64) for a in (lon, lat, elev, lst, y))
    v = np.zeros(3)
    rc = lib().orc_get_vario_params(C.c_int(lon.size), _ptr(lon, _dp), _ptr(lat, _dp), _ptr(elev, _dp),
                                    _ptr(lst, _dp), _ptr(y, _dp), C.c_double(max_ngh_dist), _ptr(v, _dp))
    return rc, v


def build_krig_params(db, prm, pt, mth):
    v = np.zeros(3)
    used = C.c_int()
    rc = lib().orc_build_krig_params(C.byref(db.c), C.byref(prm), C.byref(pt), C.c_int(mth), _ptr(v, _dp),
                                     C.byref(used))
    return rc, v, used.value


def krigall(db, prm, pt, nnghs, excl=-1, rm_zero_dist=False):
    norms = np.zeros(12)
    vario = np.zeros((12, 3))
    rc = lib().orc_krigall(C.byref(db.c), C.byref(prm), C.byref(pt), C.c_int(nnghs), C.c_int32(excl),
                           C.c_int(int(rm_zero_dist)), _ptr(norms, _dp), _ptr(vario, _dp))
    return rc, norms, vario


# ---- second tier (8f-3): monthly / annual aggregation --------------------------------------------
_AGG_DT = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def agg_groups(day_year, day_month):
    day_year = np.ascontiguousarray(day_year, np.int32)
    day_month = np.ascontiguousarray(day_month, np.int32)
    grp = np.zeros(day_year.size, np.int32)
    nyr, nmth = C.c_int32(), C.c_int32()
    rc = lib().orc_agg_groups(C.c_int64(day_year.size), _ptr(day_year, _ip), _ptr(day_month, _ip),
                              C.byref(nyr), C.byref(nmth), _ptr(grp, _ip))
    return rc, nyr.value, nmth.value, grp


def daily_to_mthly(daily, day_group, ng):
    """daily: [ndays, ...] int16 raw / f4 / f8 (NaN = masked) -> f8 [ng, ...] (NaN = masked)."""
    daily = np.ascontiguousarray(daily)
    nd = daily.shape[0]
    ncell = int(np.prod(daily.shape[1:], dtype=np.int64))
    out = np.empty((ng,) + daily.shape[1:], np.float64)
    lib().orc_daily_to_mthly(C.c_void_p(daily.ctypes.data), C.c_int(_AGG_DT[daily.dtype]), C.c_int64(nd),
                             C.c_int64(ncell), _ptr(np.ascontiguousarray(day_group, np.int32), _ip),
                             C.c_int(ng), _ptr(out, _dp))
    return out


def mthly_to_ann(mthly, nyr, nmth):
    mthly = np.ascontiguousarray(mthly, np.float64)
    ncell = int(np.prod(mthly.shape[1:], dtype=np.int64))
    out = np.empty((nyr,) + mthly.shape[1:], np.float64)
    lib().orc_mthly_to_ann(_ptr(mthly, _dp), C.c_int(nyr), C.c_int(nmth), C.c_int64(ncell), _ptr(out, _dp))
    return out


def pack_mthly_i16(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(x.shape, np.int16)
    lib().orc_pack_mthly_i16(_ptr(x, _dp), C.c_int64(x.size), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


# ---- second tier (8f-4): point-mode predictor sampling --------------------------------------------
def sample_points(lons, lats, data, lon, lat, order=0, missing=-9999.0):
    """Raster [nrows, ncols] north-up f4 (NaN = missing) sampled at the points; returns
    (val, row, col, status)."""
    lons, lats = np.ascontiguousarray(lons, np.float64), np.ascontiguousarray(lats, np.float64)
    data = np.ascontiguousarray(data, np.float32)
    lon, lat = np.atleast_1d(np.asarray(lon, np.float64)), np.atleast_1d(np.asarray(lat, np.float64))
    n = lon.size
    val, row, col, st = np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    L = lib()
    v, r, c = C.c_double(), C.c_int32(), C.c_int32()
    for i in range(n):
        st[i] = L.orc_sample_point(C.c_int(lats.size), C.c_int(lons.size), _ptr(lons, _dp), _ptr(lats, _dp),
                                   _ptr(data, _fp), C.c_double(lon[i]), C.c_double(lat[i]), C.c_int(order),
                                   C.c_double(missing), C.byref(v), C.byref(r), C.byref(c))
        val[i], row[i], col[i] = v.value, r.value, c.value
    return val, row, col, st
