"""The tie guard (include/twx.h: TWX_FLAG_NO_TIE_GUARD; twx_daily.h: note_day; twx_hip.hip: run_tie_guard).

tmin_tmax_fixer's test ``tmin >= tmax`` (interp_tair.py:170) is a discontinuity: a day whose Tmax - Tmin lies within the
fast covariance build's ~1e-6 degC of 0 can fall on the other side of it than in the reference's fp64 arithmetic, and
then the day moves by degrees, the recomputed normals by ~0.05 degC and ninvalid by 1.  The library re-kriges every
cell that has a day with |Tmax - Tmin| < 2e-5 degC on the fp64 covariance build and recomputes its whole series.

Here such days are PLANTED: the Tmax observations of one day are shifted until, in the oracle's fp64 evaluation, the
chosen cell's Tmax - Tmin is +-5e-8 degC (the GWR hat row sums to 1, so a common shift of the day's observations moves
the cell's value by the shift; single f4 ulps of ONE neighbour's observation then steer it in steps of z_j * 1e-6)."""
import numpy as np
import pytest

CELLS = ((62, 14), (63, 17), (65, 12), (66, 20), (68, 15), (69, 22))       # inside the window the tests run
TARGETS = (5e-8, -5e-8, 8e-8, -8e-8, 2e-8, -2e-8)                           # Tmax - Tmin wanted on the planted day (degC)
DAYS = (40, 200, 410, 600, 777, 1001)                                       # chronological day index, one per cell
ROWS, COLS = slice(60, 70), slice(11, 24)


def _oracle_pair(orc, dbn, dbx, prm, grid, r, c):
    ptn = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c])
    ptx = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_day"][:, r, c])
    rn, dn, _, _ = orc.interp(dbn, prm, ptn)
    rx, dx, nx, _ = orc.interp(dbx, prm, ptx)
    assert rn == 0 and rx == 0
    return dn, dx, ptx, nx


def plant_ties(orc, grid, tmin, tmax, prm):
    """-> a copy of the Tmax database whose observations put the oracle's Tmax - Tmin of cell CELLS[i] on day DAYS[i]
    within 1.5e-8 degC of TARGETS[i]; also the achieved differences."""
    from topowx_amd import stationdb as sdb
    obs = np.array(tmax.var, np.float32, copy=True)
    good = np.nonzero(np.isnan(tmax.stns[sdb.BAD]))[0]              # oracle / library station index -> column of obs
    dbn = orc.Db(tmin)
    achieved = []
    for (r, c), target, d in zip(CELLS, TARGETS, DAYS):
        for it in range(40):
            dbx = orc.Db(sdb.StationDataWrkChk(tmax.stns, "tmax", tmax.days, obs))
            dn, dx, ptx, nx = _oracle_pair(orc, dbn, dbx, prm, grid, r, c)
            err = (dx[d] - dn[d]) - target
            if abs(err) < 1.5e-8:
                break
            if abs(err) > 3e-6:                                      # coarse: the whole day (sum of the hat row = 1)
                obs[d, :] = (obs[d, :].astype(np.float64) - err).astype(np.float32)
                continue
            # fine: f4 ulps of one neighbour's observation; a neighbour with a hat-row entry of 0.02 .. 0.2
            from topowx_amd.dates import MONTH
            m = int(tmax.days[MONTH][d])
            rc, _, k, z, idx = orc.gwr_mth(dbx, prm, ptx, float(nx[m - 1]), m)
            assert rc == 0
            cand = np.nonzero((np.abs(z) > 0.02) & (np.abs(z) < 0.2))[0]
            j = cand[it % cand.size]
            col = good[idx[j]]
            ulp = float(np.spacing(np.abs(obs[d, col])))
            steps = int(np.rint(-err / (z[j] * ulp)))
            if steps == 0:
                j = cand[(it + 1) % cand.size]
                col = good[idx[j]]
                ulp = float(np.spacing(np.abs(obs[d, col])))
                steps = int(np.sign(-err / z[j]))
            obs[d, col] = np.float32(obs[d, col] + np.float32(steps * ulp))
        else:
            raise AssertionError("could not plant a tie at cell %s" % ((r, c),))
        achieved.append(err + target)
    return sdb.StationDataWrkChk(tmax.stns, "tmax", tmax.days, obs), np.array(achieved)


@pytest.fixture(scope="module")
def planted(orc, golden_case):
    grid, tmin, tmax = golden_case
    prm = orc.params()
    tmax2, achieved = plant_ties(orc, grid, tmin, tmax, prm)
    assert np.all(np.abs(achieved - np.array(TARGETS)) < 1.5e-8)
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax2), prm, grid, daily=True, nthreads=8, rows=ROWS, cols=COLS)
    return grid, tmin, tmax2, want


def test_planting_reaches_the_oracle(planted, orc):
    """CPU: on its planted day every cell is invalid / valid in the oracle exactly as the sign of its target says, by a
    margin the fast covariance build cannot resolve (the common shift of a planted day moves the other cells too: by
    tenths of a degree, either way)."""
    grid, tmin, tmax, want = planted
    dbn, dbx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
    for (r, c), t, d in zip(CELLS, TARGETS, DAYS):
        dn, dx, _, _ = _oracle_pair(orc, dbn, dbx, prm, grid, r, c)
        assert abs((dx[d] - dn[d]) - t) < 1.5e-8
        assert want["ninvalid"][r - ROWS.start, c - COLS.start] == int((dn >= dx).sum())
        assert (dn[d] >= dx[d]) == (t <= 0)


def _run(lib, grid, tmin, tmax, flags):
    ctx = lib.Context(flags=flags)
    ctx.set_stations(lib.TMIN, tmin)
    ctx.set_stations(lib.TMAX, tmax)
    got = ctx.interp_grid(grid, daily=True, rows=ROWS, cols=COLS)
    t = ctx.timing()
    ctx.close()
    return got, t


@pytest.mark.gpu
def test_planted_ties_are_decided_as_the_oracle_decides_them(planted):
    from topowx_amd import _lib as lib
    grid, tmin, tmax, want = planted
    got, t = _run(lib, grid, tmin, tmax, 0)
    assert np.array_equal(got["status"], want["status"]) and np.all(got["status"] == 0)
    assert np.array_equal(got["ninvalid"], want["ninvalid"])
    assert t["tie_cells"] >= len(CELLS) and t["tie_solves"] == 24 * t["tie_cells"]
    for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax"):
        assert np.abs(got[k].astype(np.float64) - want[k]).max() < 1e-4, k
    for k in ("daily_tmin", "daily_tmax"):
        assert np.abs(got[k].astype(int) - want[k].astype(int)).max() <= 1
    # the guarded cells carry the fp64 build's results, bit for bit: everything of theirs equals a TWX_FLAG_UK_F64_ALL run
    exact, te = _run(lib, grid, tmin, tmax, lib.FLAG_UK_F64_ALL)
    assert te["tie_cells"] == 0
    for (r, c) in CELLS:
        i, j = r - ROWS.start, c - COLS.start
        for k in ("norm_tmin", "norm_tmax", "se_tmin", "se_tmax", "daily_tmin", "daily_tmax"):
            assert np.array_equal(got[k][:, i, j], exact[k][:, i, j]), (k, r, c)
            assert np.array_equal(got[k][:, i, j], want[k][:, i, j]) or k.startswith(("norm", "se")), (k, r, c)
    # ... and the whole window of the fp64 run is the oracle's in every integer
    for k in ("daily_tmin", "daily_tmax", "ninvalid"):
        assert np.array_equal(exact[k], want[k]), k


@pytest.mark.gpu
def test_guard_off_is_the_round5_behaviour_and_no_host_sync_guards_too(planted):
    from topowx_amd import _lib as lib
    grid, tmin, tmax, want = planted
    off, t = _run(lib, grid, tmin, tmax, lib.FLAG_NO_TIE_GUARD)
    assert t["tie_cells"] == 0 and t["tie_solves"] == 0
    # without the guard a planted day is decided by the fast build's ~1e-6 degC: ninvalid may differ there -- and only there
    d = off["ninvalid"] != want["ninvalid"]
    planted_mask = np.zeros_like(d)
    for (r, c) in CELLS:
        planted_mask[r - ROWS.start, c - COLS.start] = True
    assert not np.any(d & ~planted_mask)
    nosync, t2 = _run(lib, grid, tmin, tmax, lib.FLAG_NO_HOST_SYNC)
    assert np.array_equal(nosync["ninvalid"], want["ninvalid"]) and t2["tie_cells"] >= len(CELLS)


@pytest.mark.gpu
def test_point_mode_guards_the_same_ties(planted):
    """PtInterpTair.interp_pt (the reference's per-point contract, interp_tair.py:526-592) on the planted cells: the facade
    sees the near tie in the two series and interpolates the point once more in the exact precision."""
    from topowx_amd import stationdb as sdb
    from topowx_amd.interp import PtInterpTair
    grid, tmin, tmax, want = planted
    p = PtInterpTair(tmin, tmax)
    try:
        for (r, c), t in zip(CELLS, TARGETS):
            pt = p.a_pt
            pt[sdb.LON], pt[sdb.LAT], pt[sdb.ELEV], pt[sdb.TDI] = grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c]
            for m in range(1, 13):
                pt["tmin%02d" % m], pt["tmax%02d" % m] = grid["lst_night"][m - 1, r, c], grid["lst_day"][m - 1, r, c]
            tmin_d, tmax_d, nmin, nmax, smin, smax, ninv = p.interp_pt()
            i, j = r - ROWS.start, c - COLS.start
            assert ninv == want["ninvalid"][i, j], (r, c, t)
            assert np.abs(nmin - want["norm_tmin"][:, i, j]).max() < 1e-4 and np.abs(nmax - want["norm_tmax"][:, i, j]).max() < 1e-4
            assert np.all(tmin_d < tmax_d)
    finally:
        p.close()
