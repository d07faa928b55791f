"""Randomised GPU-vs-oracle checks of the small entry points (seeded): fixer + normals recompute on random series,
aggregation on random date ranges and shapes, int16 packing on random values."""
import datetime as dt

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fix_pair_random_series(orc):
    from topowx_amd import _lib
    from topowx_amd.dates import MONTH, YEAR, get_days_metadata
    days = get_days_metadata(dt.date(1979, 6, 1), dt.date(1984, 3, 31))
    rng = np.random.default_rng(11)
    ns = 40
    t = np.arange(days.size)
    tmin = 5 - 10 * np.cos(2 * np.pi * t / 365.25)[None, :] + rng.normal(0, 3, (ns, days.size))
    gap = np.abs(rng.normal(6, 4, (ns, days.size))) * np.sign(rng.random((ns, days.size)) - rng.uniform(0, 0.2, (ns, 1)))
    tmax = tmin + gap                                                  # 0..20 % of the days inverted, per series
    ctx = _lib.Context()
    ctx.set_days(days)
    fa, fb, ninv, nmin, nmax, st = ctx.fix_pair(tmin, tmax)
    ctx.close()
    for i in range(ns):
        rc, wmin, wmax, wn = orc.fixer(tmin[i], tmax[i])
        assert st[i] == (0 if rc == 0 else 5) and (rc != 0 or ninv[i] == wn)
        if rc == 0:
            np.testing.assert_array_equal(fa[i], wmin)                 # same sequential arithmetic: bit-exact
            np.testing.assert_array_equal(fb[i], wmax)
            if wn > 0:
                np.testing.assert_allclose(nmin[i], orc.recompute_norms(wmin, days[MONTH], days[YEAR]), rtol=0, atol=1e-12)
                np.testing.assert_allclose(nmax[i], orc.recompute_norms(wmax, days[MONTH], days[YEAR]), rtol=0, atol=1e-12)


@pytest.mark.parametrize("seed", range(4))
def test_aggregate_random_axes(orc, seed):
    from topowx_amd import _lib
    from topowx_amd.dates import MONTH, YEAR, get_days_metadata
    rng = np.random.default_rng(100 + seed)
    d0 = dt.date(1950 + int(rng.integers(0, 60)), int(rng.integers(1, 13)), int(rng.integers(1, 28)))
    days = get_days_metadata(d0, d0 + dt.timedelta(days=int(rng.integers(20, 1500))))
    shape = (int(rng.integers(1, 40)), int(rng.integers(1, 70)))
    raw = rng.integers(-5000, 5000, (days.size,) + shape).astype(np.int16)
    raw[rng.random(raw.shape) < 0.03] = -32767
    rc, nyr, nmth, grp = orc.agg_groups(days[YEAR], days[MONTH])
    want = orc.daily_to_mthly(raw, grp, nyr * nmth)
    ctx = _lib.Context()
    ctx.set_days(days)
    got = ctx.aggregate(raw, mthly=True, mthly_i16=True, ann=True)
    ctx.close()
    np.testing.assert_array_equal(got["mthly"], want)
    np.testing.assert_array_equal(got["mthly_i16"], orc.pack_mthly_i16(want))
    np.testing.assert_array_equal(got["ann"], orc.mthly_to_ann(want, nyr, nmth))


def test_pack_random_values(orc):
    from topowx_amd import _lib
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.normal(0, 25, 200000), np.round(rng.normal(0, 25, 50000), 2) + 0.005,   # rounding ties
                        np.array([0.0, -0.0, 0.005, -0.005, 327.66, -327.67])])
    ctx = _lib.Context()
    got = ctx.pack_i16(x)
    ctx.close()
    np.testing.assert_array_equal(got, orc.pack_i16(x))
