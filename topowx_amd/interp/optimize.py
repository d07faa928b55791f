"""Cross-validation callers of the same kernels (twx/interp/optimize.py:376-604)."""
import numpy as np

from .. import _lib
from ..stationdb import BAD, ELEV, LAT, LON, STN_ID, TDI, get_lst_varname, get_norm_varname
from .station_select import raise_for_status

__all__ = ["build_nstn_bandwidths", "XvalTairOverall", "XvalTairAnom", "XvalTairNorm", "StationKrigParams"]


def build_nstn_bandwidths(rng_min, rng_max, pct_step):
    """optimize.py:376-405: ladder of station bandwidths (35..147 for 35, 150, 0.10)."""
    out, n = [], rng_min
    while n <= rng_max:
        out.append(n)
        n = n + np.round(pct_step * n)
    return np.array(out, dtype=np.int64)


class _XvalBase(object):
    def __init__(self, stn_da, tair_var, device=0):
        self.stn_da = stn_da
        self.var = _lib.TMAX if tair_var == "tmax" else _lib.TMIN
        self.ctx = _lib.Context(device=device)
        self.ctx.set_stations(self.var, stn_da)
        good = np.isnan(stn_da.stns[BAD])
        self.stns = stn_da.stns[good]
        self._obs_cols = np.nonzero(good)[0]
        self.idx = {s: i for i, s in enumerate(self.stns[STN_ID])}
        self.mth_masks = stn_da.mth_idx

    def _pts(self, ids):
        j = np.array([self.idx[s] for s in ids])
        st = self.stns[j]
        lst = np.column_stack([st[get_lst_varname(m)] for m in range(1, 13)])
        return j, self.ctx.make_pts(st[LON], st[LAT], st[ELEV], st[TDI], lst)

    def close(self):
        self.ctx.close()


class XvalTairOverall(_XvalBase):
    """Leave-one-out interpolation of normals + daily values (optimize.py:548-604):
    StationSelect(rm_zero_dist_stns=True) and stns_rm = the station's own id."""

    def run_interp(self, stn_id):
        d, n, s = self.run_interp_many([stn_id])
        return d[0], n[0], s[0]

    def run_interp_many(self, stn_ids, daily=True, raise_on_error=True):
        """Batched form.  ``raise_on_error=False`` is the step24 worker's behaviour (step24:52-62): a station that
        cannot be interpolated keeps NaN outputs and its TWX_CELL_* code is returned as a fourth array."""
        j, pts = self._pts(stn_ids)
        d, norms, se, st = self.ctx.interp_points(self.var, pts, excl=j, rm_zero_dist=True, daily=daily)
        if not raise_on_error:
            return d, norms, se, st
        for s in st:
            raise_for_status(s)
        return d, norms, se


class XvalTairAnom(_XvalBase):
    """Leave-one-out GWR anomalies over a ladder of bandwidths (optimize.py:476-545).
    Returns bias, MAE and r^2, each [n_bandwidths, 12] as the reference does (:510-545); the statistics are
    reduced on the device (``twx_gwr_xval_points``), only three numbers per (bandwidth, month) come back."""

    def run_xval(self, stn_id, a_nnghs):
        bias, mae, r2 = self.run_xval_many([stn_id], a_nnghs)
        return bias[0], mae[0], r2[0]

    def run_xval_many(self, stn_ids, a_nnghs, raise_on_error=True):
        """Batched form: arrays [n_stations, n_bandwidths, 12].  ``raise_on_error=False`` is the step23 worker's
        behaviour (step23:56-66): a station for which any (bandwidth, month) fails is reported in a fourth
        boolean array instead of raising."""
        a_nnghs = np.asarray(a_nnghs, np.int32)
        j, pt = self._pts(stn_ids)
        ns, nb = len(stn_ids), a_nnghs.size
        # one GPU point per (station, bandwidth, month)
        pts = np.repeat(pt, nb * 12)
        mth = np.tile(np.arange(1, 13, dtype=np.int32), ns * nb)
        nn = np.tile(np.repeat(a_nnghs, 12), ns)
        own = np.repeat(j.astype(np.int32), nb * 12)
        norm = np.column_stack([self.stns[j][get_norm_varname(m)] for m in range(1, 13)])      # [ns, 12]
        pn = np.repeat(norm, nb, axis=0).reshape(-1)                                           # (station, bw, month)
        bias, mae, r2, _, st = self.ctx.gwr_xval_points(self.var, pts, pn, mth, nn, own, own, rm_zero_dist=True)
        ok = (st.reshape(ns, nb * 12) == 0).all(axis=1)
        if raise_on_error:
            for q in st:
                raise_for_status(q)
        out = tuple(a.reshape(ns, nb, 12) for a in (bias, mae, r2))
        return out if raise_on_error else out + (ok,)


class XvalTairNorm(_XvalBase):
    """Leave-one-out xval of the normals over a ladder of bandwidths with variogram fitting
    (optimize.py:209-266, step21).  Returns err[12, n_bandwidths] = interpolated - observed."""

    def run_xval(self, stn_id, abw_nngh):
        return self.run_xval_many([stn_id], abw_nngh)[0]

    def run_xval_many(self, stn_ids, abw_nngh, raise_on_error=True):
        """Batched form: err[n_stations, 12, n_bandwidths].  ``raise_on_error=False`` (the step21 worker,
        step21:55-62) also returns ok[n_stations]: False where any (bandwidth, month) could not be solved."""
        abw = np.asarray(abw_nngh, np.int32)
        j, pt = self._pts(stn_ids)
        ns, nb = len(stn_ids), abw.size
        # one GPU point per (station, bandwidth, month)
        pts = np.repeat(pt, nb * 12)
        mth = np.tile(np.arange(1, 13, dtype=np.int32), ns * nb)
        nn = np.tile(np.repeat(abw, 12), ns)
        excl = np.repeat(j.astype(np.int32), nb * 12)
        # variogram fit + kriging with the fitted model per (station, bandwidth, month): ONE call (KrigTairAll.krigall's shape,
        # interp_tair.py:722-769): one station selection and one set of pair distances serve both stages
        mean, _, _, _, st = self.ctx.krigall_points(self.var, pts, mth, nnghs=nn, excl=excl, rm_zero_dist=True)
        if raise_on_error:
            for q in st:
                raise_for_status(q)
        obs = np.column_stack([self.stns[j][get_norm_varname(m)] for m in range(1, 13)])      # [ns, 12]
        interp = mean.reshape(ns, nb, 12)
        err = np.transpose(interp - obs[:, None, :], (0, 2, 1))                                 # [ns, 12, nb]
        if raise_on_error:
            return err
        ok = (st == 0).reshape(ns, nb * 12).all(axis=1)
        return err, ok


class StationKrigParams(_XvalBase):
    """Per-station variogram parameters for every month (optimize.py:408-474, step22).  As in the
    reference the station stays inside its own neighbourhood (rm_zero_dist_stns=False, no stns_rm)."""

    def get_krig_params(self, stn_id):
        nug, psill, rng = self.get_krig_params_many([stn_id])
        return nug[0], psill[0], rng[0]

    def get_krig_params_many(self, stn_ids, raise_on_error=True):
        """Batched form: (nug, psill, rng), each [n_stations, 12].  ``raise_on_error=False`` (the step22 worker,
        step22:52-62: a station whose fit fails gets the f8 fill value for all twelve months) also returns
        ok[n_stations]."""
        j, pt = self._pts(stn_ids)
        pts = np.repeat(pt, 12)
        mth = np.tile(np.arange(1, 13, dtype=np.int32), len(stn_ids))
        vario, _, st = self.ctx.fit_vario_points(self.var, pts, mth)
        if raise_on_error:
            for q in st:
                raise_for_status(q)
        v = vario.reshape(len(stn_ids), 12, 3)
        if raise_on_error:
            return v[:, :, 0], v[:, :, 1], v[:, :, 2]
        return v[:, :, 0], v[:, :, 1], v[:, :, 2], (st == 0).reshape(len(stn_ids), 12).all(axis=1)
