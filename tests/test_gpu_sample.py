"""GPU parity of the point-mode predictor sampling through the C-ABI: order 0 against
the executed ``GeoNc.get_row_col`` fixture, order 0 / 1 against the oracle, and
``PtInterpTair.interp_to_lonlat`` against ``interp_pt`` on the hand-filled point."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_sample_v1.npz"))


@pytest.fixture(scope="module")
def ctx():
    from topowx_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def test_get_row_col_golden(ctx, gold):
    data = np.arange(gold["lat"].size * gold["lon"].size, dtype=np.float32).reshape(gold["lat"].size, -1)
    val, row, col, st = ctx.sample_points(gold["lon"], gold["lat"], data, gold["qlon"], gold["qlat"], order=0)
    inside = gold["row"] >= 0
    np.testing.assert_array_equal(st == 0, inside)
    np.testing.assert_array_equal(row[inside], gold["row"][inside])
    np.testing.assert_array_equal(col[inside], gold["col"][inside])
    np.testing.assert_array_equal(val[inside], data[gold["row"][inside], gold["col"][inside]])


@pytest.mark.parametrize("order", [0, 1])
def test_vs_oracle(ctx, orc, gold, order):
    lon, lat = gold["lon"], gold["lat"]
    rng = np.random.default_rng(9)
    data = rng.normal(0, 50, (lat.size, lon.size)).astype(np.float32)
    data[rng.random(data.shape) < 0.05] = np.nan
    qx = rng.uniform(lon[0] - 0.01, lon[-1] + 0.01, 5000)
    qy = rng.uniform(lat[-1] - 0.01, lat[0] + 0.01, 5000)
    qx[:lon.size], qy[:lon.size] = lon, lat[5]                       # exactly on cell centres
    want = orc.sample_points(lon, lat, data, qx, qy, order=order, missing=-9999.0)
    got = ctx.sample_points(lon, lat, data, qx, qy, order=order, missing=-9999.0)
    np.testing.assert_array_equal(got[3], want[3])
    ok = want[3] == 0
    np.testing.assert_array_equal(got[0][ok], want[0][ok])           # same fp64 expression: bit-exact
    if order == 0:
        np.testing.assert_array_equal(got[1][ok], want[1][ok])
        np.testing.assert_array_equal(got[2][ok], want[2][ok])


def test_interp_to_lonlat(golden_case):
    from topowx_amd.interp import PtInterpTair
    from topowx_amd.stationdb import ELEV, LAT, LON, MASK, TDI
    import make_golden as mg
    grid, stn_tmin, stn_tmax = mg.case_inputs()
    rasters = {ELEV: dict(lon=grid["lon"], lat=grid["lat"], data=grid["elev"]),
               TDI: dict(lon=grid["lon"], lat=grid["lat"], data=grid["tdi"]),
               MASK: dict(lon=grid["lon"], lat=grid["lat"], data=grid["mask"].astype(np.float32))}
    for m in range(12):
        rasters["tmin%02d" % (m + 1)] = dict(lon=grid["lon"], lat=grid["lat"], data=grid["lst_night"][m])
        rasters["tmax%02d" % (m + 1)] = dict(lon=grid["lon"], lat=grid["lat"], data=grid["lst_day"][m])
    p = PtInterpTair(stn_tmin, stn_tmax, aux_fpaths=rasters)
    r, c = np.argwhere(grid["mask"])[7]
    lon = grid["lon"][c] + 0.3 * (grid["lon"][1] - grid["lon"][0])   # inside cell (r, c), off-centre
    lat = grid["lat"][r] - 0.2 * abs(grid["lat"][1] - grid["lat"][0])
    got = p.interp_to_lonlat(lon, lat)
    assert p.a_pt[LON] == grid["lon"][c] and p.a_pt[LAT] == grid["lat"][r]      # snapped (chgLatLon)
    assert p.a_pt[ELEV] == grid["elev"][r, c] and p.a_pt[TDI] == grid["tdi"][r, c] and p.a_pt[MASK] == 1
    want = p.interp_pt()
    for a, b in zip(got[:6], want[:6]):
        np.testing.assert_array_equal(a, b)
    # a masked cell is refused as in the reference
    off = np.argwhere(np.asarray(grid["mask"]) == 0)
    if off.size:
        r0, c0 = off[0]
        with pytest.raises(Exception, match="outside interpolation region"):
            p.interp_to_lonlat(grid["lon"][c0], grid["lat"][r0])
    p.close()
