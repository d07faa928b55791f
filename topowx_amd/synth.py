"""Deterministic synthetic inputs for the interpolation hot path.

The reference ships no sample data, so parity tests and the
benchmark run on synthetic grids / station databases whose shapes follow
SURVEY.md section 8(d): 30-arcsec cell centres with a north-up latitude axis
(``step25:113-116``), analytic elevation / TDI / LST predictor fields, a station
table carrying every per-station field the path reads and an
optional float32 observation matrix ``[ndays, nstns]``.

Everything is a pure function of the seed (``numpy.random.default_rng``).
"""
import datetime as _dt
import os

import numpy as np

from . import stationdb as sdb
from .dates import MONTH, get_days_metadata

CELL = 1.0 / 120.0  # 30 arc-seconds

# optimize.py:376-405 build_nstn_bandwidths(35, 150, 0.10) (step21:198)
NNGH_LADDER = np.array([35, 39, 43, 47, 52, 57, 63, 69, 76, 84, 92, 101, 111,
                        122, 134, 147])

CONFIGS = {
    # name: (nrows, ncols, lat_north, lon_west, nstns, seed)
    "C1": (100, 100, 46.0, -111.0, 500, 0),
    "C2": (250, 250, 46.0, -111.0, 10000, 1),
    "C3": (3250, 7000, 51.6, -125.0, 12000, 2),
}


def _fbm(lon, lat, seed):
    """Smooth analytic pseudo-terrain in about [-1, 1]."""
    r = np.random.default_rng(1000 + seed)
    out = np.zeros(np.broadcast(lon, lat).shape)
    amp, tot = 1.0, 0.0
    for octave in range(5):
        f = 0.9 * 2.0 ** octave
        ph = r.uniform(0, 2 * np.pi, 4)
        ang = r.uniform(0, np.pi)
        u = np.cos(ang) * lon + np.sin(ang) * lat
        v = -np.sin(ang) * lon + np.cos(ang) * lat
        out = out + amp * (np.sin(f * u + ph[0]) * np.cos(f * v + ph[1])
                           + 0.5 * np.sin(1.7 * f * v + ph[2]) * np.sin(1.3 * f * u + ph[3]))
        tot += 1.5 * amp
        amp *= 0.55
    return out / tot * 1.6


def field_elev(lon, lat):
    return 1500.0 + 1200.0 * _fbm(lon, lat, 0)


def field_tdi(lon, lat):
    return np.clip(50.0 + 60.0 * _fbm(lon, lat, 1), 0.0, 100.0)


def field_climdiv(lon, lat):
    """Integer id of the 1x1 degree block (so every tile sees >= 1 region)."""
    return (np.floor(lat) + 90.0) * 360.0 + (np.floor(lon) + 180.0)


def _lst_seasonal(day):
    m = np.arange(1, 13)
    base, amp = (22.0, 13.0) if day else (8.0, 11.0)
    return base + amp * np.cos(2 * np.pi * (m - 7) / 12.0)


def field_lst(lon, lat, elev, day, noise):
    """[12, ...] land-skin temperature: a_m - 0.006 elev + 3 sin(lat) + noise."""
    a = _lst_seasonal(day).reshape((12,) + (1,) * np.ndim(elev))
    return a - 0.006 * elev + 3.0 * np.sin(np.deg2rad(lat)) + noise


def make_grid(config="C1", nrows=None, ncols=None, lat_north=None, lon_west=None,
              seed=None, full_mask=None):
    """Predictor grid for one BASELINE.json config (or a custom extent).

    Returns a dict with ``lat[Y]`` (descending), ``lon[X]``, ``mask[Y,X]`` u8,
    ``elev``/``tdi`` f4, ``climdiv`` i4, ``lst_night``/``lst_day`` f4 ``[12,Y,X]``
    -- the native-dtype planes the coordinator slices in ``tiling.py:190-213``.
    """
    c = CONFIGS[config]
    nrows = c[0] if nrows is None else nrows
    ncols = c[1] if ncols is None else ncols
    lat_north = c[2] if lat_north is None else lat_north
    lon_west = c[3] if lon_west is None else lon_west
    seed = c[5] if seed is None else seed
    lat = lat_north - (np.arange(nrows) + 0.5) * CELL
    lon = lon_west + (np.arange(ncols) + 0.5) * CELL
    rng = np.random.default_rng(7000 + seed)
    if full_mask is None:
        full_mask = config != "C3"
    # the random streams are drawn whole and in order (the values do not depend on how the fields are evaluated) ...
    noise_n = rng.standard_normal((12, nrows, ncols))
    noise_d = rng.standard_normal((12, nrows, ncols))
    elev = np.empty((nrows, ncols), np.float32)
    tdi = np.empty((nrows, ncols), np.float32)
    climdiv = np.empty((nrows, ncols), np.int32)
    lst_n = np.empty((12, nrows, ncols), np.float32)
    lst_d = np.empty((12, nrows, ncols), np.float32)
    blob = None if full_mask else np.empty((nrows, ncols))

    # ... the analytic fields are elementwise: row bands on threads (numpy releases the GIL; the full configs[2] grid
    # -- 22.75 M cells -- takes 110 s on one core)
    def band(sl):
        lon2, lat2 = np.meshgrid(lon, lat[sl])
        elev[sl] = field_elev(lon2, lat2).astype(np.float32)
        tdi[sl] = field_tdi(lon2, lat2).astype(np.float32)
        climdiv[sl] = field_climdiv(lon2, lat2).astype(np.int32)
        e64 = elev[sl].astype(np.float64)
        lst_n[:, sl] = field_lst(lon2, lat2, e64, False, noise_n[:, sl]).astype(np.float32)
        lst_d[:, sl] = field_lst(lon2, lat2, e64, True, noise_d[:, sl]).astype(np.float32)
        if blob is not None:
            blob[sl] = _fbm(lon2 * 0.35, lat2 * 0.35, 50 + 7)

    nthr = 1 if nrows * ncols < (1 << 20) else min(16, os.cpu_count() or 1)
    step = -(-nrows // (4 * nthr)) if nthr > 1 else nrows
    bands = [slice(i, min(nrows, i + step)) for i in range(0, nrows, step)]
    if nthr == 1:
        for sl in bands:
            band(sl)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(band, bands))
    del noise_n, noise_d
    mask = np.ones((nrows, ncols), np.uint8) if full_mask else \
        (blob > np.quantile(blob[::8, ::8], 0.43)).astype(np.uint8)
    return dict(lat=lat, lon=lon, mask=mask, elev=elev, tdi=tdi, climdiv=climdiv,
                lst_night=lst_n, lst_day=lst_d,
                bbox=(lat.min(), lat.max(), lon.min(), lon.max()))


def make_mask(config="C3", nrows=None, ncols=None, lat_north=None, lon_west=None):
    """The land mask ``make_grid(config, ..., full_mask=False)`` would carry, without the predictor planes (the tile
    deal of the full configs[2] grid needs only this: 3250x7000 cells in ~2 s)."""
    c = CONFIGS[config]
    nrows = c[0] if nrows is None else nrows
    ncols = c[1] if ncols is None else ncols
    lat = (c[2] if lat_north is None else lat_north) - (np.arange(nrows) + 0.5) * CELL
    lon = (c[3] if lon_west is None else lon_west) + (np.arange(ncols) + 0.5) * CELL
    blob = np.empty((nrows, ncols))

    def band(sl):
        lon2, lat2 = np.meshgrid(lon, lat[sl])
        blob[sl] = _fbm(lon2 * 0.35, lat2 * 0.35, 50 + 7)

    nthr = 1 if nrows * ncols < (1 << 20) else min(16, os.cpu_count() or 1)
    step = -(-nrows // (4 * nthr)) if nthr > 1 else nrows
    bands = [slice(i, min(nrows, i + step)) for i in range(0, nrows, step)]
    if nthr == 1:
        for sl in bands:
            band(sl)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(band, bands))
    return (blob > np.quantile(blob[::8, ::8], 0.43)).astype(np.uint8)


def _km(lon, lat):
    return np.stack([lon * 78.6, lat * 111.2], axis=-1)


def _rff_exp_field(xy_km, length_km, nfeat, rng):
    """Random-Fourier-feature GP draw with an exponential covariance."""
    z = rng.standard_normal((nfeat, 2))
    w = rng.standard_normal((nfeat, 1))
    omega = z / np.abs(w) / length_km
    b = rng.uniform(0, 2 * np.pi, nfeat)
    return np.sqrt(2.0 / nfeat) * np.cos(xy_km @ omega.T + b).sum(axis=1)


def make_stations(bbox, nstns, seed, var="tmin", days=None, with_obs=False,
                  nan_frac=0.02, expand_deg=2.0, dup_frac=0.0):
    """Serially-complete station DB for one variable.

    ``bbox`` = (min lat, max lat, min lon, max lon) of the grid; stations are
    uniform in the bbox expanded by ``expand_deg``.  ``dup_frac`` > 0 puts that
    fraction of stations exactly on grid-cell centres is NOT done here; see
    tests for coincident-point cases.
    """
    is_tmax = var == "tmax"
    rng = np.random.default_rng(seed * 2 + (1 if is_tmax else 0) + 100)
    lat0, lat1, lon0, lon1 = bbox
    lon = np.round(rng.uniform(lon0 - expand_deg, lon1 + expand_deg, nstns), 5)
    lat = np.round(rng.uniform(lat0 - expand_deg, lat1 + expand_deg, nstns), 5)
    # unique coordinates (duplicates make the kriging matrix singular, step20:51-57)
    _, first = np.unique(np.stack([lon, lat], 1), axis=0, return_index=True)
    keep = np.sort(first)
    lon, lat = lon[keep], lat[keep]
    n = lon.size
    stns = np.zeros(n, dtype=sdb.stn_dtype())
    stns[sdb.STN_ID] = ["S%07d" % i for i in range(n)]
    stns[sdb.LON], stns[sdb.LAT] = lon, lat
    elev = field_elev(lon, lat) + rng.normal(0, 40.0, n)
    stns[sdb.ELEV] = elev
    stns[sdb.TDI] = np.clip(field_tdi(lon, lat) + rng.normal(0, 5.0, n), 0, 100)
    stns[sdb.MASK] = 1.0
    stns[sdb.BAD] = np.nan
    cdiv = field_climdiv(lon, lat)
    stns[sdb.CLIMDIV] = cdiv
    lst = field_lst(lon, lat, elev, is_tmax, rng.standard_normal((12, n)))
    xy = _km(lon, lat)
    udiv, inv = np.unique(cdiv, return_inverse=True)
    off = 9.0 if is_tmax else 0.0
    for m in range(1, 13):
        stns[sdb.get_lst_varname(m)] = lst[m - 1]
        gp = _rff_exp_field(xy, 40.0, 256, rng)
        stns[sdb.get_norm_varname(m)] = (12.0 + off - 0.0065 * elev + 0.35 * lst[m - 1]
                                         + gp + rng.normal(0, 0.3, n))
        stns[sdb.get_optim_varname(m)] = rng.choice(NNGH_LADDER, udiv.size)[inv]
        stns[sdb.get_optim_anom_varname(m)] = rng.choice(NNGH_LADDER, udiv.size)[inv]
        stns[sdb.get_krigparam_varname(m, sdb.VARIO_NUG)] = rng.uniform(0.1, 0.6, n)
        stns[sdb.get_krigparam_varname(m, sdb.VARIO_PSILL)] = rng.uniform(0.2, 2.0, n)
        stns[sdb.get_krigparam_varname(m, sdb.VARIO_RNG)] = rng.uniform(10.0, 80.0, n)
    # out-of-domain stations: good neighbours whose optim/vario fields are NaN
    # (interp_tair.py:489-492; SURVEY.md Appendix A closing note)
    nan_stn = rng.random(n) < nan_frac
    stns[sdb.MASK][nan_stn] = np.nan
    for m in range(1, 13):
        for nm in (sdb.get_optim_varname(m), sdb.get_optim_anom_varname(m),
                   sdb.get_krigparam_varname(m, sdb.VARIO_NUG),
                   sdb.get_krigparam_varname(m, sdb.VARIO_PSILL),
                   sdb.get_krigparam_varname(m, sdb.VARIO_RNG)):
            stns[nm][nan_stn] = np.nan

    obs = None
    if days is None:
        days = get_days_metadata(_dt.date(1981, 1, 1), _dt.date(1981, 12, 31))
    if with_obs:
        obs = make_obs(stns, days, seed, is_tmax)
    return sdb.StationDataWrkChk(stns, var, days, obs)


def make_obs(stns, days, seed, is_tmax, chunk=2048):
    """obs[d, j] = norm_m(d),j + AR(1) regional anomaly (rho .7, sigma 4, ~200 km) + N(0,1)."""
    from scipy.signal import lfilter
    n, nd = stns.size, days.size
    shared = np.random.default_rng(seed + 555)       # field shared by tmin and tmax
    own = np.random.default_rng(seed * 2 + (1 if is_tmax else 0) + 777)
    xy = _km(stns[sdb.LON], stns[sdb.LAT])
    nfeat = 48

    def basis(r):
        omega = r.standard_normal((nfeat, 2)) / 200.0
        b = r.uniform(0, 2 * np.pi, nfeat)
        return (np.sqrt(2.0 / nfeat) * np.cos(xy @ omega.T + b)).astype(np.float32)

    def ar1(r):
        e = r.standard_normal((nd, nfeat)) * np.sqrt(1 - 0.7 ** 2)
        return lfilter([1.0], [1.0, -0.7], e, axis=0).astype(np.float32)

    phi_s, c_s = basis(shared), ar1(shared)
    phi_o, c_o = basis(own), ar1(own)
    a_s, a_o = (0.8, 0.6) if is_tmax else (1.0, 0.0)
    norms = np.stack([stns[sdb.get_norm_varname(m)] for m in range(1, 13)]).astype(np.float32)
    mth = days[MONTH] - 1
    obs = np.empty((nd, n), np.float32)
    for s in range(0, nd, chunk):
        e = min(nd, s + chunk)
        anom = 4.0 * (a_s * (c_s[s:e] @ phi_s.T) + a_o * (c_o[s:e] @ phi_o.T))
        obs[s:e] = norms[mth[s:e]] + anom + own.standard_normal((e - s, n)).astype(np.float32)
    return obs


def make_case(config="C1", with_obs=False, days=None, nstns=None, **grid_kw):
    """(grid, stn_da_tmin, stn_da_tmax) for a BASELINE.json config."""
    grid = make_grid(config, **grid_kw)
    c = CONFIGS[config]
    nstns = c[4] if nstns is None else nstns
    tmin = make_stations(grid["bbox"], nstns, c[5], "tmin", days, with_obs)
    tmax = make_stations(grid["bbox"], nstns, c[5], "tmax", days, with_obs)
    return grid, tmin, tmax
