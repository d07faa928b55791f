"""Size-independent properties at BASELINE.json's full C2 size (one 250 x 250 tile, 10 000 stations, 12 monthly
Tmin normals + SE through the C-ABI): determinism, window consistency, exactness on station cells, and the oracle
on a sampled set of cells spread over the tile (the oracle cannot do 62 500 cells in seconds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    from topowx_amd import _lib, synth
    grid = synth.make_grid("C2")
    stn = synth.make_stations(grid["bbox"], synth.CONFIGS["C2"][4], synth.CONFIGS["C2"][5], "tmin")
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, stn, with_obs=False)
    full = ctx.interp_grid(grid, variables=("tmin",))
    yield dict(lib=_lib, grid=grid, stn=stn, ctx=ctx, full=full)
    ctx.close()


def test_full_tile_all_cells_ok_and_deterministic(c2):
    full, ctx, grid = c2["full"], c2["ctx"], c2["grid"]
    assert full["status"].shape == (250, 250) and np.all(full["status"] == 0)
    assert np.isfinite(full["norm_tmin"]).all() and (full["se_tmin"] > 0).all()
    again = ctx.interp_grid(grid, variables=("tmin",))
    for k in ("norm_tmin", "se_tmin", "status"):
        assert np.array_equal(full[k], again[k]), k                     # bit-identical run to run


def test_windows_equal_the_full_tile(c2):
    """A window is interpolated with other tile candidate lists, batches and launch shapes; every cell must still
    get the same neighbours and the same solve."""
    full, ctx, grid = c2["full"], c2["ctx"], c2["grid"]
    for rs, cs in ((slice(0, 64), slice(0, 64)), (slice(93, 157), slice(181, 250)), (slice(249, 250), slice(0, 250))):
        win = ctx.interp_grid(grid, variables=("tmin",), rows=rs, cols=cs)
        assert np.array_equal(win["status"], full["status"][rs, cs])
        for k in ("norm_tmin", "se_tmin"):
            assert np.array_equal(win[k], full[k][:, rs, cs]), (k, rs, cs)


def test_sampled_cells_vs_oracle(c2, orc):
    full, grid, stn = c2["full"], c2["grid"], c2["stn"]
    db, prm = orc.Db(stn), orc.params()
    rng = np.random.default_rng(7)
    cells = np.column_stack([rng.integers(0, 250, 60), rng.integers(0, 250, 60)])
    worst = 0.0
    for r, c in cells:
        pt = orc.make_pt(grid["lon"][c], grid["lat"][r], grid["elev"][r, c], grid["tdi"][r, c], grid["lst_night"][:, r, c])
        rc, _, norms, se = orc.interp(db, prm, pt, daily=False)
        assert rc == 0
        worst = max(worst, np.abs(full["norm_tmin"][:, r, c] - norms).max(), np.abs(full["se_tmin"][:, r, c] - se).max())
    assert worst < 1e-4, worst                                           # north_star tolerance (degC)


def test_daily_tile_windows_and_sampled_cells(orc):
    """The daily path on the full 250 x 250 tile with 10 000 stations (three years of days to keep the oracle side
    short): every cell done, a window equals the full tile bit for bit (int16 days, ninvalid), sampled cells match
    the oracle to the packing's last bit or one unit of it."""
    import datetime as dt
    from topowx_amd import _lib, synth
    from topowx_amd.dates import get_days_metadata
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1983, 12, 31))
    grid = synth.make_grid("C2")
    tmin = synth.make_stations(grid["bbox"], 10000, 1, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], 10000, 1, "tmax", days, with_obs=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    full = ctx.interp_grid(grid, daily=True)
    assert np.all(full["status"] == 0) and full["daily_tmin"].shape == (days.size, 250, 250)
    rs, cs = slice(120, 141), slice(7, 71)
    win = ctx.interp_grid(grid, daily=True, rows=rs, cols=cs)
    ctx.close()
    for k in ("daily_tmin", "daily_tmax", "norm_tmin", "norm_tmax", "se_tmin", "se_tmax", "ninvalid"):
        assert np.array_equal(win[k], full[k][..., rs, cs]), k
    dbn, dbx, prm = orc.Db(tmin), orc.Db(tmax), orc.params()
    rng = np.random.default_rng(3)
    for r, c in zip(rng.integers(0, 250, 6), rng.integers(0, 250, 6)):
        want = orc.interp_grid(dbn, dbx, prm, grid, daily=True, nthreads=4, rows=slice(r, r + 1), cols=slice(c, c + 1))
        assert want["status"][0, 0] == 0 and want["ninvalid"][0, 0] == full["ninvalid"][r, c]
        for k in ("daily_tmin", "daily_tmax"):
            dd = np.abs(full[k][:, r, c].astype(int) - want[k][:, 0, 0].astype(int))
            assert dd.max() <= 1 and (dd == 0).mean() > 0.999, (k, r, c)
        for k in ("norm_tmin", "norm_tmax"):
            assert np.abs(full[k][:, r, c].astype(np.float64) - want[k][:, 0, 0]).max() < 1e-4
