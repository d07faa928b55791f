"""High-precision arbiter of the universal-kriging solve (TEST INFRASTRUCTURE, like the rest of oracle/).

The kriging arithmetic of the reference lives in gstat / sp (parity unpinned at that
boundary).  What CAN be pinned without R is the numerics: this module evaluates the published predictor -- the
sp / gstat WGS84 great-circle distance (Meeus, Astronomical Algorithms, ch. 11: the form sp's gcdist.c implements),
the exponential covariance, and the augmented (k + 5) universal-kriging system -- in 40-digit arithmetic (mpmath).
The fp64 oracle (orc_uk: GLS form, centred and scaled trend) and the GPU kernels (fp32 pair distances and
exponentials, fp64 factorisation, bordered Cholesky) are compared with it; the differences are pure rounding.

Known-answer distances that do NOT come from this repository's algebra (tests/test_oracle_uk.py):
  * sp's documented example  spDistsN1(cbind(c(5, 6), c(60, 60)), c(5, 60), longlat = TRUE) -> 0.00000 55.79918
  * Meeus example 11.c, Paris -> Washington: 6181.63 km
  * WGS84 arc lengths of one degree: 110.574 km (latitude, equator), 111.132 km (45 deg), 111.694 km (pole);
    111.320 km (longitude, equator), 78.847 km (45 deg)
"""
import mpmath as mp

mp.mp.dps = 40

A_KM = mp.mpf("6378.137")
F_INV = mp.mpf("298.257223563")


def ellip_dist(lon1, lat1, lon2, lat2):
    """sp / gstat great-circle distance on the WGS84 ellipsoid (km), 40 digits."""
    lon1, lat1, lon2, lat2 = (mp.mpf(float(v)) for v in (lon1, lat1, lon2, lat2))
    if lat1 == lat2 and lon1 == lon2:
        return mp.mpf(0)
    r = mp.pi / 180
    F, G, L = (lat1 + lat2) * r / 2, (lat1 - lat2) * r / 2, (lon1 - lon2) * r / 2
    sG, cG, sF, cF, sL, cL = (mp.sin(G) ** 2, mp.cos(G) ** 2, mp.sin(F) ** 2, mp.cos(F) ** 2, mp.sin(L) ** 2,
                              mp.cos(L) ** 2)
    S = sG * cL + cF * sL
    C = cG * cL + sF * sL
    w = mp.atan(mp.sqrt(S / C))
    R = mp.sqrt(S * C) / w
    D = 2 * w * A_KM
    H1 = (3 * R - 1) / (2 * C)
    H2 = (3 * R + 1) / (2 * S)
    f = 1 / F_INV
    return D * (1 + f * H1 * sF * cG - f * H2 * cF * sG)


def uk(lon, lat, elev, lst, y, pt, nug, psill, rng):
    """Universal kriging with trend 1 + lon + lat + elev + lst and covariance psill exp(-h / rng) (+ nug at h = 0;
    pure nugget for rng = 0) through the augmented system [[C, X], [X', 0]] [lam; mu] = [c0; x0] in 40 digits.
    pt = (lon, lat, elev, lst).  Returns (mean, variance) as Python floats."""
    k = len(lon)
    nug, psill, rng = mp.mpf(float(nug)), mp.mpf(float(psill)), mp.mpf(float(rng))
    c00 = nug + psill

    def cov(h):
        if h == 0:
            return c00
        return mp.mpf(0) if rng == 0 else psill * mp.exp(-h / rng)
    n = k + 5
    A = mp.zeros(n, n)
    b = mp.zeros(n, 1)
    X = [[mp.mpf(1), mp.mpf(float(lon[i])), mp.mpf(float(lat[i])), mp.mpf(float(elev[i])), mp.mpf(float(lst[i]))]
         for i in range(k)]
    x0 = [mp.mpf(1)] + [mp.mpf(float(v)) for v in pt]
    for i in range(k):
        A[i, i] = c00
        for j in range(i):
            A[i, j] = A[j, i] = cov(ellip_dist(lon[i], lat[i], lon[j], lat[j]))
        for q in range(5):
            A[i, k + q] = A[k + q, i] = X[i][q]
        b[i] = cov(ellip_dist(pt[0], pt[1], lon[i], lat[i]))
    for q in range(5):
        b[k + q] = x0[q]
    sol = mp.lu_solve(A, b)
    mean = sum(sol[i] * mp.mpf(float(y[i])) for i in range(k))
    var = c00 - sum(sol[i] * b[i] for i in range(n))
    return float(mean), float(var)
