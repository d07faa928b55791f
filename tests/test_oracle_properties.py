"""Property tests of the oracle's integer / packing / aggregation pieces (hypothesis, CPU only): invariants the
reference code has by construction, checked on random inputs."""
import numpy as np
from hypothesis import given, settings, strategies as st


@settings(max_examples=60, deadline=None)
@given(st.lists(st.floats(-60.0, 60.0, allow_nan=False, width=64), min_size=1, max_size=200))
def test_pack_i16_matches_numpy_expression(orc, xs):
    x = np.array(xs, np.float64)
    want = (np.round(x, 2) / np.float32(0.01)).astype(np.int16)          # step25:163-164
    np.testing.assert_array_equal(orc.pack_i16(x), want)


@settings(max_examples=40, deadline=None)
@given(st.integers(40, 400), st.integers(0, 2 ** 31 - 1))
def test_fixer_postconditions(orc, n, seed):
    rng = np.random.default_rng(seed)
    tmin = rng.normal(0, 5, n)
    tmax = tmin + rng.normal(4, 3, n)                                     # some days come out inverted
    bad = tmin >= tmax
    rc, fmin, fmax, ninv = orc.fixer(tmin, tmax)
    if rc != 0:                                                           # 'No valid tmin/tmax in window'
        return
    assert ninv == int(bad.sum())
    assert np.all(fmin < fmax)                                            # every day valid afterwards
    np.testing.assert_array_equal(fmin[~bad], tmin[~bad])                 # valid days untouched
    np.testing.assert_array_equal(fmax[~bad], tmax[~bad])
    np.testing.assert_allclose((fmin + fmax)[bad], (tmin + tmax)[bad], rtol=0, atol=1e-12)   # tavg preserved


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 3), st.integers(1, 12), st.integers(0, 2 ** 31 - 1))
def test_aggregation_of_constant_months_is_exact(orc, nyr, nm, seed):
    """Days that carry their own (year, month) code aggregate to exactly that code, masked days are ignored, and
    the packed monthly value is the code itself."""
    rng = np.random.default_rng(seed)
    yrs = np.repeat(np.arange(2000, 2000 + nyr), nm * 5)
    mths = np.tile(np.repeat(np.arange(1, nm + 1), 5), nyr)
    raw = (100 * ((yrs - 2000) * 12 + mths)).astype(np.int16)[:, None] * np.ones((1, 7), np.int16)
    raw[rng.random(raw.shape) < 0.15] = -32767
    rc, ny, nmo, grp = orc.agg_groups(yrs, mths)
    assert rc == 0 and (ny, nmo) == (nyr, nm)
    m = orc.daily_to_mthly(raw, grp, ny * nmo)
    code = np.array([(y * 12 + mm) for y in range(nyr) for mm in range(1, nm + 1)], np.float64)
    for g in range(ny * nmo):
        col_all_masked = (raw[grp == g] == -32767).all(axis=0)
        assert np.all(np.isnan(m[g][col_all_masked]))
        got = m[g][~col_all_masked]
        np.testing.assert_allclose(got, np.float64(np.float32(100 * code[g]) * np.float32(0.01)), rtol=1e-15)
    packed = orc.pack_mthly_i16(m)
    assert np.all((packed == -32767) == np.isnan(m))
    assert np.all(packed[~np.isnan(m)] == np.broadcast_to((100 * code)[:, None], m.shape)[~np.isnan(m)])
