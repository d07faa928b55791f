"""GPU parity of the monthly / annual aggregation through the C-ABI: against the
executed-reference fixtures (tests/golden/golden_agg_v1.npz), against the oracle on a larger cube, and
the mosaic packing against the oracle's step25 packing.  f8 means and int16 outputs are bit-exact."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
CASES = ("two_years", "partial", "one_month")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_agg_v1.npz"))


def _ctx(year, month):
    from topowx_amd import _lib
    ctx = _lib.Context(0)
    ctx.set_days({"MONTH": month, "YEAR": year})
    return ctx


@pytest.mark.parametrize("name", CASES)
def test_golden_raw_i16(gold, name):
    import make_golden_agg as mg
    raw = mg.case_inputs(name)[2]
    assert mg.input_hash(raw) == str(gold[name + "_hash"])
    ctx = _ctx(gold[name + "_year"], gold[name + "_month"])
    out = ctx.aggregate(raw, mthly=True, mthly_i16=True, ann=True)
    np.testing.assert_array_equal(out["mthly"], gold[name + "_mthly"])
    np.testing.assert_array_equal(out["ann"], gold[name + "_ann"])
    np.testing.assert_array_equal(out["mthly_i16"], gold[name + "_mthly_i16"])
    ctx.close()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_golden_float_inputs(gold, name, dt):
    import make_golden_agg as mg
    raw = mg.case_inputs(name)[2]
    f = np.where(raw == -32767, np.nan, raw * np.float32(0.01)).astype(dt)
    ctx = _ctx(gold[name + "_year"], gold[name + "_month"])
    np.testing.assert_array_equal(ctx.aggregate(f)["mthly"], gold[name + "_mthly_f8"])
    ctx.close()


@pytest.mark.parametrize("shape", [(64, 96), (37, 53), (1, 1)])     # vector path, scalar path, one cell
def test_vs_oracle_large(orc, shape):
    from topowx_amd.dates import MONTH, YEAR, get_days_metadata
    import datetime as dt
    days = get_days_metadata(dt.date(1990, 6, 10), dt.date(1993, 2, 3))
    rng = np.random.default_rng(5)
    raw = rng.integers(-4000, 4500, (days.size,) + shape).astype(np.int16)
    raw[rng.random(raw.shape) < 0.02] = -32767
    rc, nyr, nmth, grp = orc.agg_groups(days[YEAR], days[MONTH])
    want = orc.daily_to_mthly(raw, grp, nyr * nmth)
    ctx = _ctx(days[YEAR], days[MONTH])
    assert ctx.aggregate_dims() == (nyr, nmth)
    out = ctx.aggregate(raw, mthly=True, mthly_i16=True, ann=True)
    np.testing.assert_array_equal(out["mthly"], want)
    np.testing.assert_array_equal(out["ann"], orc.mthly_to_ann(want, nyr, nmth))
    np.testing.assert_array_equal(out["mthly_i16"], orc.pack_mthly_i16(want))
    ctx.close()


def test_facade_tair_aggregate(gold):
    import make_golden_agg as mg
    from topowx_amd.dates import get_days_metadata
    from topowx_amd.interp import TairAggregate
    import datetime as dt
    raw = mg.case_inputs("two_years")[2]
    days = get_days_metadata(dt.date(1999, 1, 1), dt.date(2000, 12, 31))
    tair = np.ma.masked_array(raw * np.float32(0.01), mask=raw == -32767)
    agg = TairAggregate(days)
    m = agg.daily_to_mthly(tair)
    np.testing.assert_array_equal(np.ma.filled(m, np.nan), gold["two_years_mthly"])
    np.testing.assert_array_equal(np.ma.filled(agg.daily_to_ann(tair), np.nan), gold["two_years_ann"])
    np.testing.assert_array_equal(np.ma.filled(agg.mthly_to_ann(m), np.nan), gold["two_years_ann"])
    np.testing.assert_array_equal(agg.daily_i16_to_mthly_i16(raw), gold["two_years_mthly_i16"])
    agg.close()


def test_mosaic_of_tiles(orc):
    from topowx_amd import _lib
    from topowx_amd.interp import TileGridInfo, TileMosaic
    from topowx_amd.step25 import TileStore
    rng = np.random.default_rng(2)
    info = TileGridInfo({}, {}, 3, None, None, 4, 6, 2, 2, 32)
    stores = {}
    for t in ("h00v00", "h01v00", "h01v01"):            # h00v01 is missing -> fill values
        s = TileStore(10, 4, 6, True)
        for v in ("tmin", "tmax"):
            s.a["norm_" + v][:] = rng.normal(5, 10, s.a["norm_" + v].shape)
            s.a["se_" + v][:] = rng.random(s.a["se_" + v].shape)
            s.a["daily_" + v][:] = rng.integers(-3000, 3000, s.a["daily_" + v].shape)
        s.a["norm_tmin"][:, 0, 0] = _lib.FILL_F4
        stores[t] = s
    mos = TileMosaic(info)
    tiles = ["h00v00", "h01v00", "h00v01", "h01v01"]
    dly = mos.create_dly_mosaic(tiles, "tmin", stores)
    assert dly.shape == (10, 8, 12)
    np.testing.assert_array_equal(dly[:, 0:4, 6:12], stores["h01v00"].a["daily_tmin"])
    assert (dly[:, 4:8, 0:6] == _lib.FILL_I2).all()
    norm, se = mos.create_normals_mosaic(tiles, "tmin", stores)
    want = orc.pack_i16(stores["h01v01"].a["norm_tmin"].astype(np.float64))
    want[:, 0, 0] = _lib.FILL_I2
    np.testing.assert_array_equal(norm[:, 4:8, 6:12], want)
    np.testing.assert_array_equal(se[:, 0:4, 0:6], orc.pack_i16(stores["h00v00"].a["se_tmin"].astype(np.float64)))
    assert (norm[:, 4:8, 0:6] == _lib.FILL_I2).all()
