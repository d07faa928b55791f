"""``StationSelect`` of the reference (twx/interp/station_select.py:29-192) on the GPU."""
import numpy as np

from .. import _lib
from ..stationdb import LAT, LON, STN_ID

__all__ = ["StationSelect"]


def raise_for_status(status):
    """Map a per-point status to the exception the reference would raise."""
    if status == 0:
        return
    msg = _lib.CELL_STATUS.get(int(status), "status %d" % status)
    if status == 1:
        raise IndexError(msg)
    if status == 4:
        raise FloatingPointError(msg)
    if status in (6, 7):
        raise ValueError(msg)
    raise Exception(msg)


class StationSelect(object):
    """Finds, selects and weights the neighbouring stations of a point.

    Same constructor and attributes as the reference class; ``stn_mask`` selects
    rows of ``stn_da.stns`` (the interpolation classes pass ``isnan(bad)``,
    interp_tair.py:483-487).  One GPU context is owned per StationSelect.
    """

    def __init__(self, stn_da, stn_mask=None, rm_zero_dist_stns=False, ctx=None, var=_lib.TMIN):
        self.stn_da = stn_da
        self.stns = stn_da.stns if stn_mask is None else stn_da.stns[stn_mask]
        self.rm_zero_dist_stns = rm_zero_dist_stns
        self._mask = np.ones(stn_da.stns.size, bool) if stn_mask is None else np.asarray(stn_mask, bool)
        self.var = var
        if ctx is None:
            ctx = _lib.Context()
            ctx.set_stations(var, _MaskedDb(stn_da, self._mask))
        self.ctx = ctx
        self._id_to_idx = {s: i for i, s in enumerate(self.stns[STN_ID])}
        self.ngh_stns = self.ngh_obs = self.ngh_dists = self.ngh_wgt = None

    def excl_index(self, stns_rm):
        """``stns_rm`` (station_select.py:74-103: a str, or a numpy array of ids removed with ``np.in1d``) as what the point
        entries take: -1 (nothing), ONE station index (a single id, what every caller on the reference's path passes), or
        a ``[1, m]`` array of indices (several ids; ids that are not in the table remove nothing, as ``np.in1d`` has it).
        More than ``_lib.MAX_EXCL`` ids in the table: ValueError (a library limit, never a silent truncation)."""
        if stns_rm is None:
            return -1
        if isinstance(stns_rm, (str, bytes)):
            ids = [stns_rm]
        elif isinstance(stns_rm, np.ndarray):
            ids = list(stns_rm.ravel())
        else:
            raise Exception("stns_rm must be str, unicode, or numpy array of str/unicode")       # station_select.py:77
        idx = sorted({self._id_to_idx[str(i)] for i in ids if str(i) in self._id_to_idx})
        if len(idx) <= 1:
            return idx[0] if idx else -1
        if len(idx) > _lib.MAX_EXCL:
            raise ValueError("stns_rm names %d stations of the table; the library excludes at most %d per point" % (len(idx), _lib.MAX_EXCL))
        return np.array([idx], np.int32)

    def set_ngh_stns(self, lat, lon, nnghs, load_obs=True, obs_mth=None, stns_rm=None):
        if nnghs >= self.stns.size:
            raise IndexError("index %d is out of bounds: only %d stations" % (nnghs, self.stns.size))  # :164
        if nnghs > _lib.MAX_NNGHS + 7:
            raise ValueError("nnghs above the supported maximum (%d)" % _lib.MAX_NNGHS)
        idx, dist, wgt, st = self.ctx.knn(self.var, [lon], [lat], int(nnghs), excl=self.excl_index(stns_rm),
                                          rm_zero_dist=self.rm_zero_dist_stns)
        raise_for_status(st[0])
        self.ngh_stns = self.stns[idx[0]]
        self.ngh_dists = dist[0]
        self.ngh_wgt = wgt[0]
        self.ngh_obs = self.stn_da.load_obs(self.ngh_stns[STN_ID], mth=obs_mth) if load_obs else None


class _MaskedDb(object):
    """View of a station DB restricted by a boolean mask (what StationSelect keeps)."""

    def __init__(self, stn_da, mask):
        stns = stn_da.stns.copy()
        from ..stationdb import BAD
        # the C ABI keeps stations with isnan(bad); encode the mask through it
        stns[BAD] = np.where(mask, np.nan, 1.0)
        self.stns = stns
        self.var = stn_da.var
        self.days = stn_da.days
