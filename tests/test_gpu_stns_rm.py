"""``stns_rm`` as an ARRAY of station ids (twx/interp/station_select.py:74-103 removes any number of them with np.in1d):
``twx_set_exclusions`` + the point entries against goldens made by executing the reference's StationSelect / KrigTair /
GwrTairAnom with id arrays (tests/golden/make_golden_rm.py -> golden_rm_v1.npz): 2 ... 8 removed stations among a point's 12
nearest, a foreign id in the list, a list combined with rm_zero_dist_stns on a station's own location."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def grm(golden_case):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_rm_v1.npz"))
    assert str(g["input_hash"]) == make_golden.input_hash(*golden_case), "regenerate tests/golden/golden_rm_v1.npz (make_golden_rm.py)"
    return g


def test_fixture_is_current_and_covers_two_to_eight_exclusions(grm):
    n = (grm["excl"] >= 0).sum(axis=1)
    assert sorted(set(n.tolist())) == [2, 3, 4, 5, 6, 7, 8] and grm["rmz"].sum() == 2
    for row, idx in zip(grm["excl"], grm["idx"]):
        assert not np.intersect1d(row[row >= 0], idx[idx >= 0]).size      # no removed station among the selected


@pytest.mark.gpu
def test_knn_with_exclusion_lists_matches_the_reference(grm, golden_case):
    from topowx_amd import _lib
    grid, tmin, _ = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    for k in (35, 100):
        for rmz in (0, 1):
            sel = np.nonzero((grm["k"] == k) & (grm["rmz"] == rmz))[0]
            idx, dist, wgt, st = ctx.knn(_lib.TMIN, grm["lon"][sel], grm["lat"][sel], k, excl=grm["excl"][sel], rm_zero_dist=bool(rmz))
            assert np.all(st == 0)
            assert np.array_equal(idx, grm["idx"][sel][:, :k])                                   # bit-exact
            np.testing.assert_allclose(dist, grm["dist"][sel][:, :k], rtol=1e-12, atol=1e-10)
            np.testing.assert_allclose(wgt, grm["wgt"][sel][:, :k], rtol=1e-10, atol=1e-12)
    # a pending list is consumed by ONE call; a list given for another number of points is refused; more than 8 are refused
    a, _, _, _ = ctx.knn(_lib.TMIN, grm["lon"][:1], grm["lat"][:1], 35)
    assert np.intersect1d(a[0], grm["excl"][0][grm["excl"][0] >= 0]).size > 0                  # (nothing excluded any more)
    more = np.ascontiguousarray(grm["excl"][:2, 1:])
    ctx._chk(ctx.lib.twx_set_exclusions(ctx.h, 2, more.shape[1], more.ctypes.data_as(_lib._ip)), "twx_set_exclusions")
    with pytest.raises(_lib.TwxError, match="another number of points"):
        ctx.knn(_lib.TMIN, grm["lon"][:1], grm["lat"][:1], 35)
    with pytest.raises(ValueError):
        ctx.knn(_lib.TMIN, grm["lon"][:1], grm["lat"][:1], 35, excl=np.arange(9)[None, :])
    assert ctx.lib.twx_set_exclusions(ctx.h, 1, 9, more.ctypes.data_as(_lib._ip)) != 0
    ctx.close()


@pytest.mark.gpu
def test_krig_and_gwr_with_exclusion_lists(grm, golden_case):
    from topowx_amd import _lib
    grid, tmin, _ = golden_case
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    r, c = grm["kr_cell"][:, 0], grm["kr_cell"][:, 1]
    pts = ctx.make_pts(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c].T)
    mean, var, used, st, _ = ctx.krig_points(_lib.TMIN, pts, grm["kr_mth"], excl=grm["kr_excl"])
    assert np.all(st == 0)
    assert np.abs(mean - grm["kr_mean"]).max() < TOL and np.abs(var - grm["kr_var"]).max() < TOL
    series, used, st = ctx.gwr_points(_lib.TMIN, pts, grm["kr_mean"], grm["kr_mth"], excl=grm["kr_excl"])
    assert np.all(st == 0)
    for i, n in enumerate(grm["gw_len"]):
        assert np.abs(series[i, :n] - grm["gw_series"][i, :n]).max() < TOL, i
    # and they DO differ from the single-exclusion result (the lists are not silently cut to their first entry)
    m1, _, _, _, _ = ctx.krig_points(_lib.TMIN, pts, grm["kr_mth"], excl=grm["kr_excl"][:, 0])
    assert np.abs(m1 - mean).max() > 1e-3
    ctx.close()


@pytest.mark.gpu
def test_facade_takes_id_arrays(grm, golden_case):
    """StationSelect.set_ngh_stns / KrigTair.krig with ``stns_rm=np.array([...ids])`` as the reference accepts them."""
    from topowx_amd import stationdb as sdb
    from topowx_amd.interp import KrigTair, StationSelect
    grid, tmin, _ = golden_case
    good = np.isnan(tmin.stns[sdb.BAD])
    ids = tmin.stns[sdb.STN_ID][good]
    slct = StationSelect(tmin, good)
    for q in (0, 2, 12):                                         # 2, 3 (+ a foreign id) and 8 removed stations
        row = grm["excl"][q]
        rm = np.array([ids[j] for j in row[row >= 0]] + (["NOT_A_STATION_ID"] if q == 2 else []))
        slct.set_ngh_stns(grm["lat"][q], grm["lon"][q], int(grm["k"][q]), load_obs=False, stns_rm=rm)
        got = np.array([np.nonzero(ids == s)[0][0] for s in slct.ngh_stns[sdb.STN_ID]])
        assert np.array_equal(got, grm["idx"][q][:grm["k"][q]])
        np.testing.assert_allclose(slct.ngh_wgt, grm["wgt"][q][:grm["k"][q]], rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError, match="at most"):
        slct.set_ngh_stns(grm["lat"][0], grm["lon"][0], 35, load_obs=False, stns_rm=ids[:9].copy())
    with pytest.raises(Exception, match="stns_rm must be"):
        slct.set_ngh_stns(grm["lat"][0], grm["lon"][0], 35, load_obs=False, stns_rm=["a", "b"])
    from topowx_amd.interp import build_empty_pt
    krig = KrigTair(slct)
    pt = build_empty_pt()
    i = 1
    r, c = grm["kr_cell"][i]
    pt[sdb.LON], pt[sdb.LAT], pt[sdb.ELEV], pt[sdb.TDI] = grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c]
    for m in range(1, 13):
        pt[sdb.get_lst_varname(m)] = grid["lst_night"][m - 1, r, c]
    row = grm["kr_excl"][i]
    mean, var = krig.krig(pt, int(grm["kr_mth"][i]), stns_rm=np.array([ids[j] for j in row[row >= 0]]))
    assert abs(mean - grm["kr_mean"][i]) < TOL and abs(var - grm["kr_var"][i]) < TOL
    slct.ctx.close()
