"""Point-mode predictor sampling: the oracle's get_row_col against the executed
``GeoNc`` slice (tests/golden/golden_sample_v1.npz), and known-answer properties of the restated
bilinear path (basemap is not available: parity unpinned for order 1)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_sample_v1.npz"))


def test_get_row_col_golden(orc, gold):
    data = np.arange(gold["lat"].size * gold["lon"].size, dtype=np.float32).reshape(gold["lat"].size, -1)
    val, row, col, st = orc.sample_points(gold["lon"], gold["lat"], data, gold["qlon"], gold["qlat"], order=0)
    inside = gold["row"] >= 0
    assert inside.sum() > 100 and (~inside).sum() > 10
    np.testing.assert_array_equal(st == 0, inside)
    np.testing.assert_array_equal(row[inside], gold["row"][inside])
    np.testing.assert_array_equal(col[inside], gold["col"][inside])
    np.testing.assert_array_equal(val[inside], data[gold["row"][inside], gold["col"][inside]])
    np.testing.assert_array_equal(gold["lon"][col[inside]], gold["glon"][inside])     # snapped point (chgLatLon)
    np.testing.assert_array_equal(gold["lat"][row[inside]], gold["glat"][inside])


def test_bilinear_properties(orc, gold):
    lon, lat = gold["lon"], gold["lat"]
    LON, LAT = np.meshgrid(lon, lat)
    plane = (3.0 + 2.0 * (LON - lon[0]) * 120 - 1.5 * (LAT - lat[-1]) * 120).astype(np.float32)
    rng = np.random.default_rng(0)
    qx, qy = rng.uniform(lon[0], lon[-1], 200), rng.uniform(lat[-1], lat[0], 200)
    val, _, _, st = orc.sample_points(lon, lat, plane, qx, qy, order=1)
    want = 3.0 + 2.0 * (qx - lon[0]) * 120 - 1.5 * (qy - lat[-1]) * 120
    assert (st == 0).all()
    np.testing.assert_allclose(val, want, atol=1e-4)                     # exact for a bilinear field (f4 data)
    # cell centres reproduce the data
    val, _, _, _ = orc.sample_points(lon, lat, plane, LON[::3, ::5].ravel(), LAT[::3, ::5].ravel(), order=1)
    np.testing.assert_allclose(val, plane[::3, ::5].ravel(), atol=1e-4)
    # a missing corner -> nearest cell; a missing nearest cell or a point off the raster -> missing value
    holed = plane.copy()
    holed[10, 10] = np.nan
    v, _, _, _ = orc.sample_points(lon, lat, holed, [lon[11] - 0.001], [lat[10] - 0.001], order=1, missing=-9999.0)
    assert v[0] == holed[10, 11]
    v, _, _, _ = orc.sample_points(lon, lat, holed, [lon[10] + 0.0001], [lat[10]], order=1, missing=-9999.0)
    assert v[0] == -9999.0
    v, _, _, _ = orc.sample_points(lon, lat, plane, [lon[0] - 1.0], [lat[0]], order=1, missing=-9999.0)
    assert v[0] == -9999.0
