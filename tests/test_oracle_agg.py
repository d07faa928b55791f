"""Oracle vs the executed reference ``_TairAggregate`` slice (fixtures from
tests/golden/make_golden_agg.py): group layout, f8 means bit-exact, packed int16 bit-exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

CASES = ("two_years", "partial", "one_month")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_agg_v1.npz"))


@pytest.fixture(scope="module")
def inputs():
    import make_golden_agg as mg
    return {name: mg.case_inputs(name)[2] for name in CASES}


@pytest.mark.parametrize("name", CASES)
def test_inputs_match_fixture(gold, inputs, name):
    import make_golden_agg as mg
    assert mg.input_hash(inputs[name]) == str(gold[name + "_hash"])


@pytest.mark.parametrize("name", CASES)
def test_groups_and_means(orc, gold, inputs, name):
    raw = inputs[name]
    rc, nyr, nmth, grp = orc.agg_groups(gold[name + "_year"], gold[name + "_month"])
    assert rc == 0
    want = gold[name + "_mthly"]
    assert nyr * nmth == want.shape[0] and nyr == gold[name + "_ann"].shape[0]
    got = orc.daily_to_mthly(raw, grp, nyr * nmth)
    np.testing.assert_array_equal(got, want)                      # NaN == masked, bit-exact f8
    np.testing.assert_array_equal(orc.mthly_to_ann(got, nyr, nmth), gold[name + "_ann"])
    np.testing.assert_array_equal(orc.pack_mthly_i16(got), gold[name + "_mthly_i16"])


@pytest.mark.parametrize("name", CASES)
def test_float_inputs(orc, gold, inputs, name):
    raw = inputs[name]
    _, nyr, nmth, grp = orc.agg_groups(gold[name + "_year"], gold[name + "_month"])
    f4 = np.where(raw == -32767, np.nan, raw * np.float32(0.01)).astype(np.float32)
    np.testing.assert_array_equal(orc.daily_to_mthly(f4, grp, nyr * nmth), gold[name + "_mthly_f8"])
    np.testing.assert_array_equal(orc.daily_to_mthly(f4.astype(np.float64), grp, nyr * nmth),
                                  gold[name + "_mthly_f8"])


def test_empty_groups_are_masked(orc, gold):
    # "partial" starts in March 2003 and ends in November 2004: Jan/Feb 2003 and Dec 2004 have no day
    m = gold["partial_mthly"]
    assert np.isnan(m[0]).all() and np.isnan(m[1]).all() and np.isnan(m[23]).all()
    assert (gold["partial_mthly_i16"][[0, 1, 23]] == -32767).all()
