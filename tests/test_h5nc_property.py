"""CPU: property test of the NetCDF-4 layer (topowx_amd/h5nc.py): random dimensions, dtypes, chunking / deflate / shuffle,
fill values and hyperslab writes go through libhdf5 and come back as a numpy mirror predicts -- within one session, after
sync, and from a fresh read-only handle.  The named tests of tests/test_ncio.py pin the file conventions; this one
sweeps the index arithmetic (start / count / squeezed axes) and the type table."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from topowx_amd import h5nc

pytestmark = pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")

DTYPES = ["i1", "u1", "i2", "u2", "i4", "u4", "i8", "f4", "f8"]


@st.composite
def cases(draw):
    nd = draw(st.integers(1, 3))
    shape = tuple(draw(st.integers(1, 7)) for _ in range(nd))
    dtype = draw(st.sampled_from(DTYPES))
    zlib = draw(st.booleans())
    chunk = tuple(draw(st.integers(1, n)) for n in shape) if (zlib or draw(st.booleans())) else None
    fill = draw(st.one_of(st.none(), st.integers(-100, 100)))
    if fill is not None and dtype.startswith("u"):
        fill = abs(fill)
    writes = []
    for _ in range(draw(st.integers(1, 4))):
        key = []
        for n in shape:
            kind = draw(st.integers(0, 2))
            if kind == 0:                                   # a single index (the axis is squeezed)
                key.append(draw(st.integers(-n, n - 1)))
            elif kind == 1:                                 # a slice with unit stride, possibly open-ended
                a = draw(st.integers(0, n - 1))
                b = draw(st.integers(a + 1, n))
                key.append(slice(a if draw(st.booleans()) else (None if a == 0 else a), b if draw(st.booleans()) else (None if b == n else b)))
            else:
                key.append(slice(None))
        writes.append((tuple(key), draw(st.integers(0, 2 ** 31 - 1))))
    return shape, dtype, zlib, draw(st.booleans()), chunk, fill, writes


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(case=cases())
def test_hyperslab_round_trips(tmp_path, case):
    shape, dtype, zlib, shuffle, chunk, fill, writes = case
    path = os.path.join(str(tmp_path), "p.nc")
    dt = np.dtype(dtype)
    ds = h5nc.Dataset(path, "w")
    names = ["d%d" % i for i in range(len(shape))]
    for nme, n in zip(names, shape):
        ds.createDimension(nme, n)
    v = ds.createVariable("v", dtype, names, zlib=zlib, shuffle=shuffle, chunksizes=chunk, fill_value=fill)
    v.units = "K"
    v.scale_factor = np.float64(0.01)
    mirror = np.full(shape, fill if fill is not None else 0, dt)
    never_written = np.ones(shape, bool)
    for key, seed in writes:
        rng = np.random.default_rng(seed)
        sub = mirror[key]
        val = rng.integers(0, 100, sub.shape).astype(dt) if dt.kind in "iu" else rng.normal(size=sub.shape).astype(dt)
        v[key] = val
        mirror[key] = val
        never_written[key] = False
        got = v[key]
        assert got.shape == np.shape(sub) and np.array_equal(got, mirror[key])
    ds.sync()
    keep = ~never_written if fill is None else np.ones(shape, bool)          # (without a fill value the unwritten part is the library's default)
    assert np.array_equal(v[...][keep], mirror[keep])
    ds.close()
    ro = h5nc.Dataset(path, "r")
    w = ro.variables["v"]
    assert w.dimensions == tuple(names) and w.shape == shape and w.dtype == dt
    assert w.units == "K" and float(w.scale_factor) == 0.01
    if fill is not None:
        assert w._FillValue == dt.type(fill)
    if chunk is not None:
        assert tuple(w.chunking()) == tuple(chunk)
        assert bool(w.filters()["zlib"]) == zlib
    full = w[...]
    assert np.array_equal(full[keep], mirror[keep])
    for key, _ in writes:
        assert np.array_equal(w[key], full[key])
    assert [ro.dimensions[n] if isinstance(ro.dimensions[n], int) else len(ro.dimensions[n]) for n in names] == list(shape)
    ro.close()


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(words=st.lists(st.text(alphabet=st.characters(min_codepoint=32, max_codepoint=0x24F), max_size=12), min_size=1, max_size=9),
       width=st.integers(1, 16))
def test_string_variables_round_trip(tmp_path, words, width):
    """Variable-length strings (station ids, names as netCDF4-python writes ``str``) and fixed-width byte strings."""
    path = os.path.join(str(tmp_path), "s.nc")
    ds = h5nc.Dataset(path, "w")
    ds.createDimension("n", len(words))
    vs = ds.createVariable("name", str, ("n",))
    fs = ds.createVariable("code", "S%d" % width, ("n",))
    vs[:] = np.array(words, dtype=object)
    raw = np.array([w.encode("ascii", "replace")[:width] for w in words], dtype="S%d" % width)
    fs[:] = raw
    ds.title = "a é b"
    ds.close()
    ro = h5nc.Dataset(path, "r")
    assert list(ro.variables["name"][:]) == words
    assert np.array_equal(ro.variables["code"][:], raw)
    assert ro.variables["name"][len(words) - 1] == words[-1]
    assert ro.title == "a é b"
    ro.close()
