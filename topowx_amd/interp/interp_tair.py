"""``twx.interp.interp_tair`` facade: KrigTair, GwrTairAnom, InterpTair, PtInterpTair.

Call signatures, return values and error behaviour follow
twx/interp/interp_tair.py of the reference; the arithmetic runs in libtwxhip.
"""
import numpy as np

from .. import _lib
from ..dates import MONTH, YEAR, get_mth_metadata
from ..stationdb import (BAD, CLIMDIV, ELEV, LAT, LON, MASK, TDI, StationDataWrkChk, get_lst_varname,
                         get_norm_varname, get_optim_anom_varname, get_optim_varname)
from .station_select import StationSelect, raise_for_status

__all__ = ["GwrTairAnom", "KrigTair", "KrigTairAll", "BuildKrigParams", "InterpTair", "StationDataWrkChk",
           "PtInterpTair", "build_empty_pt", "tmin_tmax_fixer", "chunk_to_grid"]

DFLT_INIT_NNGHS = 100  # interp_tair.py:51


def build_empty_pt():
    """interp_tair.py:200-213: the structured scalar a point is described with."""
    dt = [(LON, np.float64), (LAT, np.float64), (ELEV, np.float64), (TDI, np.float64), (CLIMDIV, np.float64),
          (MASK, np.float64)]
    dt.extend([("tmin%02d" % m, np.float64) for m in range(1, 13)])
    dt.extend([("tmax%02d" % m, np.float64) for m in range(1, 13)])
    dt.extend([(get_norm_varname(m), np.float64) for m in range(1, 13)])
    dt.extend([(get_optim_varname(m), np.float64) for m in range(1, 13)])
    dt.extend([(get_lst_varname(m), np.float64) for m in range(1, 13)])
    dt.extend([(get_optim_anom_varname(m), np.float64) for m in range(1, 13)])
    return np.zeros(1, dtype=dt)[0]


def _pt_to_twx(ctx, pt):
    return ctx.make_pts(pt[LON], pt[LAT], pt[ELEV], pt[TDI], [pt[get_lst_varname(m)] for m in range(1, 13)])


def tmin_tmax_fixer(tmin, tmax, tail=15, ctx=None):
    """interp_tair.py:143-197 on the GPU (twx_fix_pair).  Returns (tmin, tmax, ninvalid)."""
    tmin = np.asarray(tmin, np.float64)
    tmax = np.asarray(tmax, np.float64)
    own = ctx is None or ctx.ndays != tmin.size
    if own:
        from ..dates import get_days_metadata
        import datetime as dt
        ctx = _lib.Context(fixer_tail=tail)
        d0 = dt.date(2001, 1, 1)
        ctx.set_days(get_days_metadata(d0, d0 + dt.timedelta(days=int(tmin.size) - 1)))
    fa, fb, ninv, _, _, st = ctx.fix_pair(tmin, tmax)
    if own:
        ctx.close()
    raise_for_status(st[0])
    return fa[0], fb[0], int(ninv[0])


class KrigTair(object):
    """Moving-window regression kriging of monthly normals (interp_tair.py:771-926)."""

    def __init__(self, stn_slct):
        self.stn_slct = stn_slct
        self.ci_critval = -1.959963984540054  # stats.norm.ppf(0.025), interp_tair.py:792

    def std_err_ci(self, tair_mean, tair_var):
        std_err = np.sqrt(tair_var) if tair_var >= 0 else 0
        ci_r = np.abs(std_err * self.ci_critval)
        return std_err, (tair_mean - ci_r, tair_mean + ci_r)

    def krig(self, pt, mth, nnghs=None, vario_params=None, stns_rm=None):
        s = self.stn_slct
        mean, var, _, st, _ = s.ctx.krig_points(
            s.var, _pt_to_twx(s.ctx, pt), int(mth), nnghs=None if nnghs is None else int(nnghs),
            vario=None if vario_params is None else vario_params, excl=s.excl_index(stns_rm),
            rm_zero_dist=s.rm_zero_dist_stns)
        raise_for_status(st[0])
        return mean[0], var[0]


class GwrTairAnom(object):
    """GWR of daily anomalies for one month (interp_tair.py:215-314)."""

    def __init__(self, stn_slct):
        self.stn_slct = stn_slct

    def gwr_mth(self, pt, mth, nnghs=None, stns_rm=None):
        s = self.stn_slct
        out, _, st = s.ctx.gwr_points(s.var, _pt_to_twx(s.ctx, pt), pt[get_norm_varname(mth)], int(mth),
                                      nnghs=None if nnghs is None else int(nnghs), excl=s.excl_index(stns_rm),
                                      rm_zero_dist=s.rm_zero_dist_stns)
        raise_for_status(st[0])
        return out[0, :s.ctx.mth_days[mth - 1]]


class InterpTair(object):
    """Normals + daily values of one variable at a point (interp_tair.py:371-439)."""

    def __init__(self, krig_tair, gwr_tair):
        self.krig_tair = krig_tair
        self.gwr_tair = gwr_tair
        self.mth_masks = gwr_tair.stn_slct.stn_da.mth_idx
        self.ndays = gwr_tair.stn_slct.stn_da.days.size

    def interp(self, pt, stns_rm=None):
        s = self.gwr_tair.stn_slct
        daily, norms, se, st = s.ctx.interp_points(s.var, _pt_to_twx(s.ctx, pt), excl=s.excl_index(stns_rm),
                                                   rm_zero_dist=s.rm_zero_dist_stns)
        raise_for_status(st[0])
        for m in range(1, 13):          # the reference stores the normals on the point (:433)
            pt[get_norm_varname(m)] = norms[0, m - 1]
        return daily[0], norms[0], se[0]


class PredictorGrids(object):
    """Predictor rasters sampled at a point (interp_tair.py:87-141).

    ``rasters``: ordered mapping ``a_pt field -> dict(lon=[ncols], lat=[nrows] north-up, data=[nrows, ncols],
    missing_value=...)`` -- the array-level stand-in for the reference's one-variable netCDF files (the
    file reader is SURVEY.md 8f-2).  ``interp_orders[i] == 1`` marks rasters that are interpolated
    bilinearly when the point is not snapped to the grid."""

    def __init__(self, rasters, interp_orders=None, ctx=None, device=0):
        self.rasters = dict(rasters)
        orders = np.zeros(len(self.rasters)) if interp_orders is None else interp_orders
        self.orders = {k: int(o) for k, o in zip(self.rasters, orders)}
        self._own = ctx is None
        self.ctx = _lib.Context(device) if ctx is None else ctx
        self._f4 = {}
        for k, r in self.rasters.items():
            d = np.ma.filled(np.ma.masked_invalid(np.ma.asarray(r["data"], dtype=np.float64)), np.nan)
            if r.get("missing_value") is not None:
                d = np.where(d == r["missing_value"], np.nan, d)
            self._f4[k] = d.astype(np.float32)

    def close(self):
        if self._own:
            self.ctx.close()

    def _sample(self, name, lon, lat, order):
        r = self.rasters[name]
        lons = np.asarray(r["lon"], np.float64)
        lons = np.where(lons > 180, lons - 360.0, lons)                  # util_ncdf.py:266-268
        miss = r.get("missing_value")
        return self.ctx.sample_points(lons, r["lat"], self._f4[name], lon, lat, order=order,
                                      missing=np.nan if miss is None else float(miss)), lons

    def setPtValues(self, aPt, chgLatLon=True):
        chged = False
        for name in self.rasters:
            if chgLatLon or self.orders[name] != 1:
                (val, row, col, st), lons = self._sample(name, aPt[LON], aPt[LAT], 0)
                if st[0] != 0:
                    raise IndexError("point (%r, %r) is outside raster %r" % (aPt[LON], aPt[LAT], name))
                v = val[0]
                if np.isnan(v) and self.rasters[name].get("missing_value") is not None:
                    v = self.rasters[name]["missing_value"]
                aPt[name] = v
                if chgLatLon and not chged:
                    aPt[LON] = lons[col[0]]
                    aPt[LAT] = np.asarray(self.rasters[name]["lat"], np.float64)[row[0]]
                    chged = True
            else:
                (val, _, _, _), _ = self._sample(name, aPt[LON], aPt[LAT], 1)
                aPt[name] = val[0]

    def sample(self, lon, lat, order_override=None):
        """Batched form: every raster at all points -> dict of arrays (status 1 = outside, order 0)."""
        out = {}
        for name in self.rasters:
            order = self.orders[name] if order_override is None else order_override
            (val, row, col, st), _ = self._sample(name, lon, lat, order)
            out[name] = val
            out[name + "_status"] = st
        return out


def chunk_to_grid(wrk_chk):
    """The reference's f8[32, Y, X] work chunk (tiling.py:205-213) as the grid dict the C ABI bindings take."""
    w = np.asarray(wrk_chk)
    return dict(lat=w[3, :, 0], lon=w[4, 0, :], mask=(w[2] != 0) & np.isfinite(w[2]),
                elev=w[5], tdi=w[6], climdiv=np.nan_to_num(w[7]), lst_night=w[8:20], lst_day=w[20:32])


class PtInterpTair(object):
    """Tmin and Tmax at a point / over a work chunk (interp_tair.py:441-592).

    ``interp_pt`` keeps the reference's one-point-per-call contract;
    ``interp_chunk`` is the batched entry the GPU wants: it takes the reference's
    f8[32, Y, X] work chunk (tiling.py:205-213) and returns the arrays the worker
    writes (step25:163-172).
    """

    def __init__(self, stn_da_tmin, stn_da_tmax, aux_fpaths=None, interp_orders=None, norms_only=False,
                 device=0):
        self.days = stn_da_tmin.days
        self.stn_da_tmin = stn_da_tmin
        self.stn_da_tmax = stn_da_tmax
        self.norms_only = norms_only
        self.ctx = _lib.Context(device=device)
        self.ctx.set_stations(_lib.TMIN, stn_da_tmin, with_obs=not norms_only)
        self.ctx.set_stations(_lib.TMAX, stn_da_tmax, with_obs=not norms_only)
        slct_n = StationSelect(stn_da_tmin, np.isnan(stn_da_tmin.stns[BAD]), ctx=self.ctx, var=_lib.TMIN)
        slct_x = StationSelect(stn_da_tmax, np.isnan(stn_da_tmax.stns[BAD]), ctx=self.ctx, var=_lib.TMAX)
        self.interp_tmin = InterpTair(KrigTair(slct_n), GwrTairAnom(slct_n))
        self.interp_tmax = InterpTair(KrigTair(slct_x), GwrTairAnom(slct_x))
        if aux_fpaths is not None:          # here: the raster mapping PredictorGrids takes (interp_tair.py:507-508)
            self.pGrids = PredictorGrids(aux_fpaths, interp_orders, ctx=self.ctx)
        self.a_pt = build_empty_pt()

    def interp_to_lonlat(self, lon, lat, fixInvalid=True, chgLatLon=True, stns_rm=None, elev=None):
        """interp_tair.py:513-524: sample the predictor rasters at (lon, lat), then ``interp_pt``."""
        self.a_pt[LON] = lon
        self.a_pt[LAT] = lat
        self.pGrids.setPtValues(self.a_pt, chgLatLon)
        if elev is not None:
            self.a_pt[ELEV] = elev
        if self.a_pt[MASK] == 0:
            raise Exception('Point is outside interpolation region')
        return self.interp_pt(fixInvalid, stns_rm)

    TIE_EPS = 2e-5      # degC: the grid entries' tie guard (include/twx.h, TWX_FLAG_NO_TIE_GUARD), here on the host

    def _both(self, stns_rm):
        pt = self.a_pt
        out = []
        for v, itp in (("tmin", self.interp_tmin), ("tmax", self.interp_tmax)):
            for m in range(1, 13):          # interp_tair.py:560-563 / :569-572
                pt[get_lst_varname(m)] = pt["%s%02d" % (v, m)]
            s = itp.gwr_tair.stn_slct
            daily, norms, se, st = self.ctx.interp_points(s.var, _pt_to_twx(self.ctx, pt), excl=s.excl_index(stns_rm),
                                                          rm_zero_dist=s.rm_zero_dist_stns,
                                                          daily=not self.norms_only)
            raise_for_status(st[0])
            out.append((daily[0] if daily is not None else None, norms[0], se[0]))
        return out

    def interp_pt(self, fix_invalid=True, stns_rm=None):
        out = self._both(stns_rm)
        if fix_invalid and not self.norms_only and np.any(np.abs(out[1][0] - out[0][0]) < self.TIE_EPS):
            # a day whose Tmax - Tmin lies within the fast covariance build's ~1e-6 degC of 0 could fall on the other side of
            # the fixer's test (interp_tair.py:170) than in the reference's fp64 arithmetic: this point once more, every
            # kriging system on the fp64 build, so that the fixed days and ninvalid do not depend on it
            self.ctx.set_precision("exact")
            try:
                out = self._both(stns_rm)
            finally:
                self.ctx.set_precision("fast")
        (tmin_dly, tmin_norms, tmin_se), (tmax_dly, tmax_norms, tmax_se) = out
        ninvalid = 0
        if fix_invalid and not self.norms_only:
            fa, fb, ninv, nmin, nmax, st = self.ctx.fix_pair(tmin_dly, tmax_dly)
            raise_for_status(st[0])
            tmin_dly, tmax_dly, ninvalid = fa[0], fb[0], int(ninv[0])
            if ninvalid > 0:                # normals recomputed from the fixed series (:583-590)
                tmin_norms, tmax_norms = nmin[0], nmax[0]
        return tmin_dly, tmax_dly, tmin_norms, tmax_norms, tmin_se, tmax_se, ninvalid

    def interp_chunk(self, wrk_chk, daily=None):
        """All unmasked cells of a reference work chunk in one GPU call.

        ``wrk_chk`` planes: 0 row, 1 col, 2 mask, 3 lat, 4 lon, 5 elev, 6 tdi, 7 climdiv,
        8-19 LST night (tmin01..12), 20-31 LST day (tmax01..12).
        """
        daily = (not self.norms_only) if daily is None else daily
        return self.ctx.interp_grid(chunk_to_grid(wrk_chk), daily=daily)

    def close(self):
        self.ctx.close()


class BuildKrigParams(object):
    """Variogram parameters of a point's neighbourhood (interp_tair.py:612-698, step22):
    bandwidth smoothed from the neighbours' optimum, then R ``get_vario_params`` -- here
    ``twx_fit_vario_points`` (parity with gstat unpinned)."""

    def __init__(self, stn_slct):
        self.stn_slct = stn_slct

    def get_krig_params(self, pt, mth, rm_stnid=None):
        s = self.stn_slct
        # the reference never passes rm_stnid on (:667,681): the station sits in its own neighbourhood
        vario, _, st = s.ctx.fit_vario_points(s.var, _pt_to_twx(s.ctx, pt), int(mth),
                                              rm_zero_dist=s.rm_zero_dist_stns)
        raise_for_status(st[0])
        return vario[0, 0], vario[0, 1], vario[0, 2]


class KrigTairAll(object):
    """Variogram fitting + kriging in one step for a given bandwidth (interp_tair.py:700-769,
    R ``krig_all`` interp.R:148-159); used by the bandwidth optimisation of step21."""

    def __init__(self, stn_slct):
        self.stn_slct = stn_slct

    def krigall(self, pt, nnghs, stns_rm=None):
        s = self.stn_slct
        pts = np.repeat(_pt_to_twx(s.ctx, pt), 12)
        mth = np.arange(1, 13, dtype=np.int32)
        excl = s.excl_index(stns_rm)
        mean, _, _, _, st = s.ctx.krigall_points(s.var, pts, mth, nnghs=int(nnghs), excl=excl,
                                                 rm_zero_dist=s.rm_zero_dist_stns)        # fit + krige, one selection
        for q in st:
            raise_for_status(q)
        return mean
