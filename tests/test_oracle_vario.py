"""Variogram estimation + range fit of the oracle (interp.R:54-113).
Known-answer checks only: gstat cannot run here (PARITY UNPINNED)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
_dp = C.POINTER(C.c_double)


def _variogram(orc, lon, lat, e, cutoff, width=5.0):
    cap = int(np.ceil(cutoff / width)) + 3
    d, g, n = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    nb = orc.lib().orc_variogram(C.c_int(lon.size), lon.ctypes.data_as(_dp), lat.ctypes.data_as(_dp),
                                 e.ctypes.data_as(_dp), C.c_double(cutoff), C.c_double(width),
                                 d.ctypes.data_as(_dp), g.ctypes.data_as(_dp), n.ctypes.data_as(_dp))
    return d[:nb], g[:nb], n[:nb]


def test_binned_semivariogram_matches_numpy(orc):
    import make_golden
    r = np.random.default_rng(0)
    k = 60
    lon, lat = r.uniform(-111.5, -110.5, k), r.uniform(44.5, 45.5, k)
    e = r.normal(0, 1, k)
    cutoff = 70.0
    d, g, n = _variogram(orc, lon, lat, e, cutoff)
    h = make_golden.ellip_dist_np(lon[:, None], lat[:, None], lon[None, :], lat[None, :])
    iu = np.triu_indices(k, 1)
    hh, dd = h[iu], (e[:, None] - e[None, :])[iu] ** 2
    keep = hh <= cutoff
    b = np.floor(hh[keep] / 5.0).astype(int)
    want_n = np.bincount(b)
    nz = want_n > 0
    assert np.array_equal(n, want_n[nz])
    np.testing.assert_allclose(d, (np.bincount(b, hh[keep]) / np.maximum(want_n, 1))[nz], rtol=1e-12)
    np.testing.assert_allclose(g, (np.bincount(b, dd[keep]) / (2 * np.maximum(want_n, 1)))[nz], rtol=1e-12)
    assert n.sum() == keep.sum()


def _fit(orc, dist, gamma, npairs, nug, psill, r0):
    out = C.c_double()
    rc = orc.lib().orc_fit_range(C.c_int(dist.size), dist.ctypes.data_as(_dp), gamma.ctypes.data_as(_dp),
                                 npairs.ctypes.data_as(_dp), C.c_double(nug), C.c_double(psill), C.c_double(r0),
                                 C.byref(out))
    return rc, out.value


def test_range_fit_recovers_exact_model(orc):
    dist = np.arange(2.5, 100, 5.0)
    npairs = np.full(dist.size, 30.0)
    for true_r in (8.0, 25.0, 60.0):
        gamma = 0.2 + 0.9 * (1 - np.exp(-dist / true_r))
        rc, r = _fit(orc, dist, gamma, npairs, 0.2, 0.9, 0.1 * dist.max())
        assert rc == 0 and abs(r - true_r) < 1e-5 * true_r
    # weights np / h^2 (fit.method = 7): the short lags dominate the fit
    gamma = 0.2 + 0.9 * (1 - np.exp(-dist / 20.0))
    gamma[10:] += 0.3
    rc, r = _fit(orc, dist, gamma, npairs, 0.2, 0.9, 9.75)
    assert rc == 0 and 18.0 < r < 26.0


def test_unusable_fit_falls_back_to_pure_nugget(orc):
    dist = np.arange(2.5, 50, 5.0)
    rc, _ = _fit(orc, dist, np.full(dist.size, 0.5), np.full(dist.size, 10.0), 0.5, 0.0, 5.0)   # psill <= 0
    assert rc == 1
    # get_vario_params on white-noise residuals around an exact trend: total sill ~ min(gamma) -> (nug, 0, 0)
    r = np.random.default_rng(3)
    k = 80
    lon, lat = r.uniform(-111.5, -110.5, k), r.uniform(44.5, 45.5, k)
    elev = r.uniform(900, 2500, k)
    lst = -5 - 0.006 * elev + r.normal(0, 1, k)
    y = 3 + 0.2 * lon - 0.4 * lat - 0.005 * elev + 0.3 * lst          # no residual at all
    rc, v = orc.get_vario_params(lon, lat, elev, lst, y + 1e-9 * r.normal(0, 1, k), 60.0)
    assert rc == 0 and v[1] >= 0 and v[2] >= 0 and v[0] + v[1] < 1e-15


def test_get_vario_params_on_correlated_field(orc):
    """A field with an exponential covariance: the fitted structure must be spatial (range > 0) and
    the total sill near the residual variance."""
    r = np.random.default_rng(5)
    import make_golden
    k = 120
    lon, lat = r.uniform(-112, -110, k), r.uniform(44, 46, k)
    elev = r.uniform(900, 2500, k)
    lst = -5 - 0.006 * elev + r.normal(0, 1, k)
    h = make_golden.ellip_dist_np(lon[:, None], lat[:, None], lon[None, :], lat[None, :])
    Cm = 1.0 * np.exp(-h / 30.0) + 0.1 * np.eye(k)
    y = 12 - 0.0065 * elev + 0.35 * lst + np.linalg.cholesky(Cm) @ r.normal(0, 1, k)
    rc, v = orc.get_vario_params(lon, lat, elev, lst, y, 150.0)
    assert rc == 0 and v[1] > 0 and 3.0 < v[2] < 200.0 and 0.2 < v[0] + v[1] < 3.0
