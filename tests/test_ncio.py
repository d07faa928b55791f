"""netCDF containers: tile files in the layout of ``TileWriter`` (tiling.py:304-537) and the station DB
layout ``StationSerialDataDb`` reads (station_data.py:547-616), in BOTH containers -- NetCDF-4 on libhdf5
(``topowx_amd.h5nc``) and classic netCDF on scipy -- compared field by field.  CPU only."""
import datetime as dt
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from topowx_amd import h5nc, ncio, synth
from topowx_amd import stationdb as sdb
from topowx_amd.dates import get_days_metadata
from topowx_amd.interp import TileGridInfo, TileMosaic


FORMATS = [pytest.param("NETCDF4", marks=pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")),
           "NETCDF3_64BIT"]


def _info():
    lons = -110.0 + (np.arange(12) + 0.5) / 120.0
    lats = 45.0 - (np.arange(8) + 0.5) / 120.0
    return TileGridInfo({0: "h00v00", 1: "h01v00"}, {"h00v00": (0, 0), "h01v00": (0, 6)}, 2, lons, lats, 8, 6, 4, 3, 32)


@pytest.mark.parametrize("fmt", FORMATS)
def test_tile_writer_roundtrip(tmp_path, fmt):
    info = _info()
    days = get_days_metadata(dt.date(1999, 12, 30), dt.date(2000, 1, 8))
    w = ncio.TileWriter(info, str(tmp_path), format=fmt)
    rng = np.random.default_rng(0)
    blocks = {}
    for (r, c) in ((0, 0), (4, 3)):                               # two of the four chunks of the tile
        b = dict(d=rng.integers(-3000, 3000, (days.size, 4, 3)).astype(np.int16),
                 n=rng.normal(5, 8, (12, 4, 3)).astype(np.float32), s=rng.random((12, 4, 3)).astype(np.float32),
                 i=rng.integers(0, 5, (4, 3)).astype(np.int32))
        w.write_tile_chunk("h01v00", "tmin", days, r, c, b["d"], b["n"], b["s"], b["i"])
        blocks[(r, c)] = b
    t = ncio.read_tile(w.fpath("h01v00", "tmin"), "tmin")
    assert t["daily"].shape == (days.size, 8, 6) and t["daily"].dtype == np.int16
    for (r, c), b in blocks.items():
        np.testing.assert_array_equal(t["daily"][:, r:r + 4, c:c + 3], b["d"])
        np.testing.assert_array_equal(t["norm"][:, r:r + 4, c:c + 3], b["n"])
        np.testing.assert_array_equal(t["se"][:, r:r + 4, c:c + 3], b["s"])
        np.testing.assert_array_equal(t["ninvalid"][r:r + 4, c:c + 3], b["i"])
    assert (t["daily"][:, 0:4, 3:6] == ncio.FILL_I2).all() and (t["ninvalid"][4:8, 0:3] == ncio.FILL_I4).all()
    assert (t["norm"][:, 4:8, 0:3] == ncio.FILL_F4).all()
    np.testing.assert_array_equal(t["lon"], info.lons[6:12])
    np.testing.assert_array_equal(t["lat"], info.lats[0:8])
    assert t["time_units"] == "days since 1999-12-30 0:0:0"
    np.testing.assert_array_equal(t["time"], np.arange(days.size) + 0.5)
    assert ncio.file_format(w.fpath("h01v00", "tmin")) == fmt
    ds = ncio.open_dataset(w.fpath("h01v00", "tmin"), "r")
    v = ds.variables["tmin"]
    assert v.dimensions == ("time", "lat", "lon") and np.asarray(v.scale_factor).dtype == np.float32
    assert np.float32(v.scale_factor) == np.float32(0.01) and np.int16(v._FillValue) == -32767
    assert v.grid_mapping == "crs" and ds.variables["crs"].grid_mapping_name == "latitude_longitude"
    assert ds.variables["tmin_normal"].dimensions == ("time_normals", "lat", "lon")
    assert ds.variables["climatology_bounds"].shape == (12, 2) and ds.Conventions == "CF-1.6"
    if fmt == "NETCDF4":                                          # the reference's chunk shapes (tiling.py:453-486)
        assert v.chunking() == [days.size, 4, 3] and ds.variables["inconsist_tair"].chunking() == [4, 3]
        assert ds.variables["tmin_se"].chunking() == [12, 4, 3] and not v.filters()["zlib"]
    # 1981-01-01 .. 2010-02-01 relative to the first day (tiling.py:412-420)
    cb = ds.variables["climatology_bounds"][:]
    assert cb[0, 0] == (dt.date(1981, 1, 1) - dt.date(1999, 12, 30)).days
    assert cb[11, 1] == (dt.date(2011, 1, 1) - dt.date(1999, 12, 30)).days
    ds.close()


@pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")
@pytest.mark.parametrize("zl", [False, True])
def test_tile_sink_whole_tiles(tmp_path, zl):
    """``ncio.TileSink``: whole tiles into the TileWriter layout (chunks gathered and pwritten to their addresses / deflated by
    the workers), two calls in flight, files prepared ahead for a tile that never comes removed, a normals-only tile."""
    from concurrent.futures import ThreadPoolExecutor
    info = _info()
    days = get_days_metadata(dt.date(1999, 12, 30), dt.date(2000, 1, 8))
    rng = np.random.default_rng(3)

    def tile(seed):
        r = np.random.default_rng(seed)
        d = r.integers(-3000, 3000, (days.size, 8, 6)).astype(np.int16)
        return {"daily_tmin": d, "daily_tmax": (d + 800).astype(np.int16), "norm_tmin": r.normal(0, 5, (12, 8, 6)).astype("f4"),
                "norm_tmax": r.normal(9, 5, (12, 8, 6)).astype("f4"), "se_tmin": r.random((12, 8, 6)).astype("f4"),
                "se_tmax": r.random((12, 8, 6)).astype("f4"), "ninvalid": r.integers(0, 7, (8, 6)).astype("i4")}
    a0, a1 = tile(0), tile(1)
    sink = ncio.TileSink(info, str(tmp_path), days, threads=3, zlib=zl, order=[0, 1], ahead=1, verify=(0, 1))
    with ThreadPoolExecutor(2) as ex:                             # (driver.interp_tiles_streamed(writer_threads=2) does this)
        list(ex.map(lambda ka: sink(*ka), ((0, a0), (1, a1))))
    sink.close()
    assert sink.stats["tiles"] == 2 and sink.stats["verified"] == 2 and sink.stats["int16_bytes"] == 4 * a0["daily_tmin"].nbytes
    for tid, a in (("h00v00", a0), ("h01v00", a1)):
        for var in ("tmin", "tmax"):
            t = ncio.read_tile(sink.writer.fpath(tid, var), var)
            for got, key in (("daily", "daily_" + var), ("norm", "norm_" + var), ("se", "se_" + var), ("ninvalid", "ninvalid")):
                np.testing.assert_array_equal(t[got], a[key])
            ds = h5nc.Dataset(sink.writer.fpath(tid, var))
            assert ds.variables[var].chunking() == [days.size, 4, 3] and ds.variables[var].filters()["zlib"] == zl
            ds.close()
    # look-ahead: tile 1's files are prepared while tile 0 is written; tile 1 never comes -> close() leaves no empty files behind
    out2 = tmp_path / "ahead"
    sink = ncio.TileSink(info, str(out2), days, threads=2, zlib=zl, order=[0, 1], ahead=1)
    sink(0, a0)
    sink.close()
    assert sorted(os.listdir(out2)) == ["h00v00"]
    # a normals-only run hands no daily block over: the small variables are written, the daily variable keeps its fill value
    out3 = tmp_path / "normals_only"
    sink = ncio.TileSink(info, str(out3), days, threads=2, zlib=zl)
    sink(1, {k: (None if k.startswith("daily_") else v) for k, v in a1.items()})
    sink.close()
    t = ncio.read_tile(sink.writer.fpath("h01v00", "tmax"), "tmax")
    np.testing.assert_array_equal(t["norm"], a1["norm_tmax"])
    np.testing.assert_array_equal(t["ninvalid"], a1["ninvalid"])
    assert (t["daily"] == ncio.FILL_I2).all() and sink.stats["int16_bytes"] == 0


@pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")
def test_tile_sink_appends_chunks_deflated_elsewhere(tmp_path):
    """``TileSink(zlib=True)`` fed with chunk bytes that are already shuffled + deflated -- what a stream with ``deflate_chunks``
    delivers (twx_stream_deflate; here the CPU restatement of the GPU's encoder makes them): ``H5Dwrite_chunk`` appends them as
    they are, libhdf5's filter pipeline (its zlib) reads the int16 values back."""
    from oracle import deflate_oracle as dorc
    info = _info()
    days = get_days_metadata(dt.date(1999, 12, 1), dt.date(2000, 3, 31))
    rng = np.random.default_rng(5)
    base = 800 * np.sin(np.arange(days.size) / 9.0)[:, None, None] + rng.normal(0, 40, (1, 8, 6))
    arrays = {"ninvalid": rng.integers(0, 7, (8, 6)).astype("i4"), "deflate_chunks": (4, 3)}
    daily = {}
    for var, off in (("tmin", -300), ("tmax", 500)):
        daily[var] = np.rint(base + off + rng.normal(0, 15, (days.size, 8, 6))).astype(np.int16)
        daily[var][:, 0, :2] = ncio.FILL_I2
        arrays["deflated_" + var] = dorc.deflate_tile(daily[var], 4, 3)
        arrays["norm_" + var] = rng.normal(0, 5, (12, 8, 6)).astype("f4")
        arrays["se_" + var] = rng.random((12, 8, 6)).astype("f4")
    sink = ncio.TileSink(info, str(tmp_path), days, threads=2, zlib=True, verify=(1,))
    sink(1, arrays)
    sink.close()
    assert sink.stats["verified"] == 1 and sink.stats["int16_bytes"] == 2 * daily["tmin"].nbytes
    for var in ("tmin", "tmax"):
        t = ncio.read_tile(sink.writer.fpath("h01v00", var), var)
        np.testing.assert_array_equal(t["daily"], daily[var])
        np.testing.assert_array_equal(t["norm"], arrays["norm_" + var])
        ds = h5nc.Dataset(sink.writer.fpath("h01v00", var))
        f = ds.variables[var].filters()
        assert f["zlib"] and f["shuffle"] and ds.variables[var].chunking() == [days.size, 4, 3]
        ds.close()
    with pytest.raises(IOError, match="zlib=True"):
        plain = ncio.TileSink(info, str(tmp_path / "p"), days, threads=2)
        try:
            plain(1, arrays)
        finally:
            plain.close()
    with pytest.raises(IOError, match="chunk shape"):
        other = ncio.TileSink(info, str(tmp_path / "o"), days, threads=2, zlib=True)
        try:
            other(1, dict(arrays, deflate_chunks=(8, 6)))
        finally:
            other.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_tiles_to_daily_mosaic(tmp_path, fmt):
    info = _info()
    days = get_days_metadata(dt.date(2001, 1, 1), dt.date(2001, 1, 5))
    w = ncio.TileWriter(info, str(tmp_path), format=fmt)
    rng = np.random.default_rng(1)
    full = {}
    for t in ("h00v00", "h01v00"):
        for v in ("tmin", "tmax"):
            d = rng.integers(-3000, 3000, (days.size, 8, 6)).astype(np.int16)
            for r in (0, 4):
                for c in (0, 3):
                    w.write_tile_chunk(t, v, days, r, c, d[:, r:r + 4, c:c + 3], np.zeros((12, 4, 3)), np.zeros((12, 4, 3)),
                                       np.zeros((4, 3)))
            full[(t, v)] = d
    stores = ncio.read_tile_stores(str(tmp_path), ["h00v00", "h01v00", "h02v00"])
    assert sorted(stores) == ["h00v00", "h01v00"]
    mos = TileMosaic(info).create_dly_mosaic(["h00v00", "h01v00"], "tmax", stores)
    np.testing.assert_array_equal(mos, np.concatenate([full[("h00v00", "tmax")], full[("h01v00", "tmax")]], axis=2))


@pytest.mark.parametrize("fmt", FORMATS)
def test_station_db_roundtrip(tmp_path, fmt):
    days = get_days_metadata(dt.date(1980, 1, 1), dt.date(1980, 3, 31))
    grid = synth.make_grid("C1")
    db = synth.make_stations(grid["bbox"], 60, 3, "tmin", days, with_obs=True)
    p = str(tmp_path / "serial_tmin.nc")
    ncio.write_station_db(p, db, format=fmt)
    assert ncio.file_format(p) == fmt
    back = ncio.read_station_db(p, "tmin", cls=sdb.StationDataWrkChk)
    assert isinstance(back, sdb.StationDataWrkChk) and back.var_name == "tmin"
    np.testing.assert_array_equal(back.stn_ids, db.stn_ids)
    np.testing.assert_array_equal(back.days.YMD, db.days.YMD)
    for name in db.stns.dtype.names:
        if name != sdb.STN_ID:
            np.testing.assert_array_equal(back.stns[name], db.stns[name])       # NaN-aware, bit-exact f8
    np.testing.assert_array_equal(back.var, db.var)
    np.testing.assert_array_equal(back.load_obs(db.stn_ids[[3, 7]], mth=2), db.load_obs(db.stn_ids[[3, 7]], mth=2))
    # the layout _build_stn_struct reads: (station_id,) columns and the ids as variable-length strings (NetCDF-4,
    # create_db_all_stations.py:274) or a (station_id, string*) char array (classic)
    ds = ncio.open_dataset(p, "r")
    assert ds.variables[sdb.STN_ID].dimensions[0] == sdb.STN_ID
    if fmt == "NETCDF4":
        assert ds.variables[sdb.STN_ID].dimensions == (sdb.STN_ID,) and ds.variables[sdb.STN_ID].dtype == object
        assert ds.variables["tmin"].chunking() == [days.size, 50]
    else:
        assert ds.variables[sdb.STN_ID].dimensions[1].startswith("string")
        nc3 = netcdf_file(p, "r", mmap=False)                       # and scipy's own reader agrees on the container
        assert nc3.variables["tmin"].dimensions == ("time", sdb.STN_ID)
        nc3.close()
    assert ds.variables["tmin"].dimensions == ("time", sdb.STN_ID)
    ds.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_path_taking_constructors_and_field_coverage(tmp_path, capsys, fmt):
    """``StationDataWrkChk(path, 'tmin')`` / ``StationSerialDataDb(path, 'tmin')`` exactly as the reference constructs
    them (step25:53-54, optimize.py:229-234), on a database written by ``write_station_db`` in either container: every field
    ``_build_stn_struct`` / ``StationSerialDataDb.__init__`` read (station_data.py:126-183,554-616) survives."""
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1981, 2, 28))
    grid = synth.make_grid("C1", nrows=20, ncols=20)
    db = synth.make_stations(grid["bbox"], 80, 5, "tmax", days, with_obs=True)
    db.stns[sdb.BAD][3] = 1.0
    p = str(tmp_path / "serial_tmax.nc")
    ncio.write_station_db(p, db, format=fmt)
    for cls in (sdb.StationDataWrkChk, sdb.StationSerialDataDb):
        a = cls(p, "tmax")
        assert type(a) is cls and a.var_name == "tmax" and a.var.dtype == np.float32
        want_fields = [sdb.LON, sdb.LAT, sdb.ELEV, sdb.TDI, sdb.MASK, sdb.BAD, sdb.CLIMDIV] + [
            namer(m) for _, namer in sdb.MONTHLY_FIELDS for m in range(1, 13)]
        assert len(want_fields) == 7 + 7 * 12
        for f in want_fields:
            np.testing.assert_array_equal(a.stns[f], db.stns[f])              # NaN (masked) entries included
        np.testing.assert_array_equal(a.stn_ids, db.stn_ids)
        np.testing.assert_array_equal(a.var, db.var)
        np.testing.assert_array_equal(a.days.YMD, db.days.YMD)
        assert a.stn_idxs[db.stn_ids[7]] == 7 and set(a.mth_idx) == set(db.mth_idx)
    # pathlib paths, missing files, and containers that are neither
    import pathlib
    assert sdb.StationDataWrkChk(pathlib.Path(p), "tmax").stns.size == 80
    with pytest.raises(IOError, match="no such station database"):
        sdb.StationDataWrkChk(str(tmp_path / "nope.nc"), "tmax")
    junk = tmp_path / "junk.nc"
    junk.write_bytes(b"GIF89a" + b"\0" * 64)
    with pytest.raises(IOError, match="neither a classic netCDF nor a NetCDF-4"):
        sdb.StationSerialDataDb(str(junk), "tmax")
    h5 = tmp_path / "truncated_hdf5.nc"
    h5.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(IOError, match="not an HDF5 / NetCDF-4 file|libhdf5"):
        sdb.StationSerialDataDb(str(h5), "tmax")
    # the command line: what is read directly, the classic fallback, a check, a conversion to the other container
    assert ncio.main(["--convert-help"]) == 0
    out = capsys.readouterr().out
    assert "nccopy -k 64-bit-offset" in out and "string16" in out and "station_data.py" in out and "DIRECTLY" in out
    assert ncio.main(["--check", p, "tmax"]) == 0
    out = capsys.readouterr().out
    assert "missing fields: none" in out and out.startswith(fmt)
    other = [f for f in ("NETCDF4", "NETCDF3_64BIT") if f != fmt][0]
    if other == "NETCDF3_64BIT" or h5nc.available():
        q = str(tmp_path / "converted.nc")
        assert ncio.main(["--convert", p, q, "tmax", other]) == 0 and ncio.file_format(q) == other
        b = sdb.StationSerialDataDb(q, "tmax")
        for f in db.stns.dtype.names:
            np.testing.assert_array_equal(b.stns[f], db.stns[f])
        np.testing.assert_array_equal(b.var, db.var)


# ---- NetCDF-4 on libhdf5 (topowx_amd.h5nc) ---------------------------------------------------------------------------
needs_h5 = pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")


def _h5dump():
    import shutil
    for c in ("h5dump", "/opt/conda/bin/h5dump"):
        p = shutil.which(c)
        if p:
            return p
    return None


@needs_h5
def test_netcdf4_tile_is_an_hdf5_file_with_dimension_scales(tmp_path):
    """What makes an HDF5 file a NetCDF-4 file, asserted on ``h5dump -H`` of a tile (netCDF-C's libhdf5 layer): every
    dimension a DIMENSION_SCALE dataset with ``_Netcdf4Dimid``, dimensions without coordinate variable under the
    'This is a netCDF dimension but not a netCDF variable.' name, DIMENSION_LIST / REFERENCE_LIST on the variables,
    chunked storage with the declared fill value, text attributes as null-terminated fixed strings."""
    import subprocess
    exe = _h5dump()
    if exe is None:
        pytest.skip("h5dump not installed")
    info = _info()
    days = get_days_metadata(dt.date(2001, 1, 1), dt.date(2001, 1, 4))
    w = ncio.TileWriter(info, str(tmp_path), format="NETCDF4", zlib=True)
    w.write_tile_chunk("h00v00", "tmax", days, 0, 0, np.zeros((4, 4, 3), np.int16), np.ones((12, 4, 3)), np.ones((12, 4, 3)),
                       np.zeros((4, 3)))
    fp = w.fpath("h00v00", "tmax")
    assert h5nc.is_hdf5(fp)
    hdr = subprocess.run([exe, "-H", "-p", fp], capture_output=True, text=True, check=True).stdout
    flat = " ".join(hdr.split())

    def block(name):
        i = hdr.index('DATASET "%s"' % name)
        j = hdr.find("\n   DATASET ", i + 1)
        return " ".join(hdr[i:j if j > 0 else None].split())
    assert 'ATTRIBUTE "_NCProperties"' in flat and 'ATTRIBUTE "Conventions"' in flat
    for dim in ("time", "lat", "lon", "nv", "time_normals"):
        b = block(dim)
        assert 'ATTRIBUTE "CLASS"' in b and 'ATTRIBUTE "NAME"' in b and 'ATTRIBUTE "_Netcdf4Dimid"' in b, dim
    assert "H5T_IEEE_F32BE" in block("nv") and 'ATTRIBUTE "REFERENCE_LIST"' in block("nv")      # a dimension without variable
    assert "H5T_IEEE_F64LE" in block("lat") and 'ATTRIBUTE "units"' in block("lat")             # a coordinate variable
    b = block("tmax")
    assert "H5T_STD_I16LE" in b and "SIMPLE { ( 4, 8, 6 ) / ( 4, 8, 6 ) }" in b
    assert 'ATTRIBUTE "DIMENSION_LIST"' in b and "H5T_VLEN { H5T_REFERENCE { H5T_STD_REF_OBJECT }}" in b
    assert "CHUNKED ( 4, 4, 3 )" in b and "COMPRESSION DEFLATE { LEVEL 4 }" in b and "PREPROCESSING SHUFFLE" in b
    assert "VALUE -32767" in b and 'ATTRIBUTE "_FillValue"' in b and "STRPAD H5T_STR_NULLTERM" in b
    assert "DATASPACE SCALAR" in block("crs")
    ds = h5nc.Dataset(fp)
    assert [ds._dimid[d] for d in ("time", "lat", "lon", "nv", "time_normals")] == [0, 1, 2, 3, 4]
    nm = ds.variables  # the hidden machinery is not an attribute of the data model
    assert "DIMENSION_LIST" not in nm["tmax"].ncattrs() and "CLASS" not in nm["lat"].ncattrs() and "nv" not in nm
    assert ds.dimensions["nv"] == 2 and "_NCProperties" not in ds.ncattrs()
    ds.close()


@needs_h5
def test_reference_shaped_netcdf4_station_db(tmp_path):
    """A database laid out as the reference's writers lay it out (create_db_all_stations.py:262-311, station_data.py:
    300-330): variable-length string ids, string columns, f8 columns with ``_FillValue`` / ``missing_value``, a column
    created WITHOUT fill value whose never-written entries hold the default fill (read back masked -> NaN by
    netCDF4-python), an integer column, a time axis that does not start at its units' origin, zlib-chunked
    observations -- opened through the reference's constructor call."""
    days = get_days_metadata(dt.date(1948, 3, 1), dt.date(1948, 4, 9))
    rng = np.random.default_rng(5)
    n = 37
    ids = np.array(sorted("GHCN_USC%08d" % i for i in rng.choice(10 ** 6, n, replace=False)))
    p = str(tmp_path / "serial_tmin.nc")
    ds = h5nc.Dataset(p, "w")
    ds.createDimension("time", days.size)
    ds.createDimension(sdb.STN_ID, n)
    t = ds.createVariable("time", "f8", ("time",))
    t.units, t.calendar = "days since 1948-1-1 0:0:0", "standard"
    t[:] = np.arange(days.size) + 60.0                                     # 1948-03-01 (leap year)
    ds.createVariable(sdb.STN_ID, str, (sdb.STN_ID,))[:] = ids.astype(object)
    ds.createVariable("station_name", str, (sdb.STN_ID,))[:] = np.array(["NAME %d" % i for i in range(n)], object)
    ds.createVariable("state", str, (sdb.STN_ID,))[:] = np.array(["MT", "ID", "WY"], object)[rng.integers(0, 3, n)]
    lon, lat = rng.uniform(-115, -105, n), rng.uniform(42, 48, n)
    elev = rng.uniform(500, 3000, n)
    for name, a in ((sdb.LON, lon), (sdb.LAT, lat), (sdb.ELEV, elev)):
        v = ds.createVariable(name, "f8", (sdb.STN_ID,), fill_value=-9999.0)
        v.missing_value = -9999.0
        v[:] = a
    bad = ds.createVariable(sdb.BAD, "f8", (sdb.STN_ID,), fill_value=ncio.FILL_F8)        # add_stn_variable(..., fill)
    bad[4] = 1.0
    nofill = ds.createVariable("optim_nnghs01", "f8", (sdb.STN_ID,))                       # created without fill value
    nofill[0:10] = np.arange(10) + 35.0
    nofill[10:] = ncio.FILL_F8                                                             # what never-written entries hold
    cd = ds.createVariable(sdb.CLIMDIV, "i4", (sdb.STN_ID,), fill_value=ncio.FILL_I4)
    cd[:] = np.where(np.arange(n) % 5 == 0, ncio.FILL_I4, 2401 + np.arange(n) % 3)
    obs = rng.normal(0, 8, (days.size, n)).astype(np.float32)
    ov = ds.createVariable("tmin", "f4", ("time", sdb.STN_ID), chunksizes=(days.size, 8), zlib=True,
                           fill_value=ncio.FILL_F4)
    ov[:] = obs
    ds.close()

    da = sdb.StationDataWrkChk(p, "tmin")                                   # step25:53
    np.testing.assert_array_equal(da.stn_ids, ids)
    assert da.days.YMD[0] == 19480301 and da.days.YMD[-1] == 19480409 and da.days.size == days.size
    np.testing.assert_array_equal(da.stns[sdb.LON], lon)
    np.testing.assert_array_equal(da.stns[sdb.ELEV], elev)
    assert da.stns["state"].dtype.kind == "U" and set(da.stns["state"]) <= {"MT", "ID", "WY"}
    assert da.stns["station_name"][7] == "NAME 7"
    want_bad = np.full(n, np.nan)
    want_bad[4] = 1.0
    np.testing.assert_array_equal(da.stns[sdb.BAD], want_bad)               # never written -> _FillValue -> NaN
    np.testing.assert_array_equal(da.stns["optim_nnghs01"][:10], np.arange(10) + 35.0)
    assert np.isnan(da.stns["optim_nnghs01"][10:]).all()                    # default fill of a variable without _FillValue
    assert da.stns[sdb.CLIMDIV].dtype == np.float64 and np.isnan(da.stns[sdb.CLIMDIV][::5]).all()
    np.testing.assert_array_equal(da.stns[sdb.CLIMDIV][1:5], [2402, 2403, 2401, 2402])
    np.testing.assert_array_equal(da.var, obs)
    np.testing.assert_array_equal(da.load_obs(ids[[2, 9, 30]], mth=4), obs[da.mth_idx[4]][:, [2, 9, 30]])
    # hyperslabs of the chunked, compressed matrix
    ds = h5nc.Dataset(p)
    v = ds.variables["tmin"]
    assert v.filters() == {"zlib": True, "shuffle": True, "complevel": 4} and v.chunking() == [days.size, 8]
    np.testing.assert_array_equal(v[3:11, 5:29], obs[3:11, 5:29])
    np.testing.assert_array_equal(v[7], obs[7])
    np.testing.assert_array_equal(v[:, -1], obs[:, -1])
    assert v[2, 3] == obs[2, 3] and v[...].shape == obs.shape
    with pytest.raises(IndexError):
        v[::2]
    with pytest.raises(h5nc.H5Error, match="read-only"):
        v[0, 0] = 1.0
    ds.close()


@needs_h5
def test_netcdf4_and_classic_round_trips_agree_field_by_field(tmp_path):
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1981, 1, 31))
    grid = synth.make_grid("C1", nrows=20, ncols=20)
    db = synth.make_stations(grid["bbox"], 70, 9, "tmin", days, with_obs=True)
    a = str(tmp_path / "a.nc")
    b = str(tmp_path / "b.nc")
    ncio.write_station_db(a, db, format="NETCDF4", zlib=True)
    ncio.write_station_db(b, db, format="NETCDF3_64BIT")
    sa, _, da, oa = ncio.read_station_db_arrays(a, "tmin")
    sb, _, db_, ob = ncio.read_station_db_arrays(b, "tmin")
    assert sa.dtype.names == sb.dtype.names
    for f in sa.dtype.names:
        np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
    np.testing.assert_array_equal(oa, ob)
    np.testing.assert_array_equal(da.YMD, db_.YMD)
    # tiles: same arrays, same attributes, from both containers
    info = _info()
    rng = np.random.default_rng(3)
    blk = (rng.integers(-3000, 3000, (days.size, 8, 6)).astype(np.int16), rng.normal(0, 5, (12, 8, 6)).astype(np.float32),
           rng.random((12, 8, 6)).astype(np.float32), rng.integers(0, 3, (8, 6)).astype(np.int32))
    out = {}
    for fmt in ("NETCDF4", "NETCDF3_64BIT"):
        w = ncio.TileWriter(info, str(tmp_path / fmt), format=fmt)
        w.write_tile_chunk("h00v00", "tmin", days, 0, 0, *blk)
        out[fmt] = ncio.read_tile(w.fpath("h00v00", "tmin"), "tmin")
        ds = ncio.open_dataset(w.fpath("h00v00", "tmin"))
        out[fmt]["attrs"] = {k: {a_: v.getncattr(a_) for a_ in v.ncattrs()} for k, v in ds.variables.items()}
        out[fmt]["glob"] = {a_: ds.getncattr(a_) for a_ in ds.ncattrs()}
        out[fmt]["dims"] = {k: (v.dimensions, v.shape, np.dtype(v.dtype)) for k, v in ds.variables.items()}
        ds.close()
    for k in ("daily", "norm", "se", "ninvalid", "lon", "lat", "time"):
        np.testing.assert_array_equal(out["NETCDF4"][k], out["NETCDF3_64BIT"][k])
    assert out["NETCDF4"]["dims"] == out["NETCDF3_64BIT"]["dims"]
    assert out["NETCDF4"]["glob"] == out["NETCDF3_64BIT"]["glob"]
    assert set(out["NETCDF4"]["attrs"]) == set(out["NETCDF3_64BIT"]["attrs"])
    for var, at in out["NETCDF4"]["attrs"].items():
        bt = out["NETCDF3_64BIT"]["attrs"][var]
        assert set(at) == set(bt), var
        for name in at:
            assert np.array_equal(at[name], bt[name]) and type(at[name]) is type(bt[name]), (var, name)


@needs_h5
def test_h5nc_data_model_corners(tmp_path):
    """Append mode, coordinate variables created after their users, a variable named like a dimension it does not lead
    with (``_nc4_non_coord_``), 2-D char coordinate (``_Netcdf4Coordinates``), scalar / string / array attributes,
    masked writes, plain-HDF5 datasets without scales (``phony_dim``)."""
    p = str(tmp_path / "m.nc")
    ds = h5nc.Dataset(p, "w")
    ds.createDimension("x", 3)
    ds.createDimension("y", 2)
    ds.createDimension("string4", 4)
    u = ds.createVariable("u", "f4", ("y", "x"), fill_value=np.float32(-1))       # before its coordinate variables
    u[:] = np.ma.masked_array(np.arange(6, dtype=np.float32).reshape(2, 3), mask=[[0, 1, 0], [0, 0, 1]])
    xv = ds.createVariable("x", "i4", ("x",))
    xv[:] = [10, 20, 30]
    yv = ds.createVariable("y", "S1", ("y", "string4"))                           # char-array coordinate
    yv[:] = np.array([list("ab\0\0"), list("cdef")], "S1")
    odd = ds.createVariable("string4", "f8", ("x",))                              # named like a dimension it does not use
    odd[:] = [1.5, 2.5, 3.5]
    ds.history = "h"
    ds.setncattr("levels", np.array([1, 2, 3], np.int16))
    ds.empty = ""
    u.valid_range = np.array([0.0, 9.0])
    ds.close()
    ds = h5nc.Dataset(p, "a")
    assert dict(ds.dimensions) == {"x": 3, "y": 2, "string4": 4}
    assert ds.variables["u"].dimensions == ("y", "x") and ds.variables["y"].dimensions == ("y", "string4")
    assert ds.variables["string4"].dimensions == ("x",)
    np.testing.assert_array_equal(ds.variables["u"][:], [[0, -1, 2], [3, 4, -1]])
    np.testing.assert_array_equal(ds.variables["string4"][:], [1.5, 2.5, 3.5])
    assert b"".join(ds.variables["y"][1]) == b"cdef"
    assert ds.history == "h" and ds.empty == "" and ds.levels.dtype == np.int16 and list(ds.levels) == [1, 2, 3]
    np.testing.assert_array_equal(ds.variables["u"].valid_range, [0.0, 9.0])
    w = ds.createVariable("w", "f8", ("x",))                                      # a new variable in append mode
    w[1:] = [7.0, 8.0]
    ds.history = "h2"                                                             # attribute overwrite
    with pytest.raises(h5nc.H5Error):
        ds.createVariable("w", "f8", ("x",))
    with pytest.raises(KeyError):
        ds.createVariable("q", "f8", ("nodim",))
    ds.close()
    ds = h5nc.Dataset(p)
    assert ds.history == "h2" and ds.variables["w"].dimensions == ("x",)
    np.testing.assert_array_equal(ds.variables["w"][1:], [7.0, 8.0])
    with pytest.raises(AttributeError):
        ds.variables["w"].getncattr("nope")
    ds.close()
    exe = _h5dump()
    if exe:
        import subprocess
        hdr = subprocess.run([exe, "-H", p], capture_output=True, text=True, check=True).stdout
        assert 'DATASET "_nc4_non_coord_string4"' in hdr and 'ATTRIBUTE "_Netcdf4Coordinates"' in hdr
    # a dataset of a plain HDF5 file (no dimension scales): anonymous dimensions by length
    lib, _ = h5nc._need()
    fid = lib.H5Fcreate(str(tmp_path / "plain.h5").encode(), 2, 0, 0)
    sp = lib.H5Screate_simple(2, h5nc._dims(2, [3, 5]), None)
    did = lib.H5Dcreate2(fid, b"grid", h5nc._g("H5T_STD_I32BE_g"), sp, 0, 0, 0)   # big-endian on disk
    a = np.arange(15, dtype=np.int32)
    lib.H5Dwrite(did, h5nc._g("H5T_NATIVE_INT_g"), 0, 0, 0, a.ctypes.data)
    lib.H5Dclose(did), lib.H5Sclose(sp), lib.H5Fclose(fid)
    ds = h5nc.Dataset(str(tmp_path / "plain.h5"))
    g = ds.variables["grid"]
    assert g.dimensions == ("phony_dim_0", "phony_dim_1") and g.dtype == np.int32
    np.testing.assert_array_equal(g[:], a.reshape(3, 5))
    ds.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_product_files_have_the_reference_layout(tmp_path, fmt):
    """The mosaic / monthly containers of tiling.py:650-736,815-925,996-1078 (creation only: the arithmetic that fills
    them runs on the GPU, tests/test_gpu_agg.py / test_gpu_facade.py)."""
    days = get_days_metadata(dt.date(1950, 1, 1), dt.date(1950, 12, 31))
    lon = -110.0 + (np.arange(12) + 0.5) / 120.0
    lat = 45.0 - (np.arange(8) + 0.5) / 120.0
    ds = ncio.create_dly_mosaic_ds(str(tmp_path / "tmin_1950.nc"), "tmin", days, lon, lat, "1.3.0", format=fmt)
    assert ds.variables["tmin"].shape == (365, 8, 12) and ds.variables["time"].units == "days since 1948-1-1 0:0:0"
    assert ds.variables["time"][0] == 731.5 and ds.variables["time_bnds"][364, 1] == 1096.0
    assert ds.title == "Daily Interpolated Topoclimatic Temperature 19500101-19501231" and "1.3.0" in ds.history
    if fmt == "NETCDF4":
        assert ds.variables["tmin"].chunking() == [1, 8, 12] and ds.variables["tmin"].filters()["zlib"]
    ds.variables["tmin"][10, 2:4, 3:5] = np.int16(123)
    ds.sync()
    mth = ncio.create_ds_mthly(ds, str(tmp_path / "tmin_1950_mthly.nc"), 1950, "tmin", "1.3.0", format=fmt)
    assert mth.variables["tmin"].shape == (12, 8, 12) and mth.variables["time"][0] == 731 + 15
    np.testing.assert_array_equal(mth.variables["time_bnds"][1], [731 + 31, 731 + 59])
    assert mth.variables["tmin"].cell_methods.startswith("time: minimum within days")
    assert np.float32(mth.variables["tmin"].scale_factor) == np.float32(0.01)
    np.testing.assert_array_equal(mth.variables["lon"][:], lon)
    assert mth.variables["crs"].grid_mapping_name == "latitude_longitude" and mth.title.endswith("for 1950")
    mth.close()
    ds.close()
    back = ncio.open_dataset(str(tmp_path / "tmin_1950.nc"))
    d = back.variables["tmin"][9:12, 1:5, 2:6]
    assert (d[1, 1:3, 1:3] == 123).all() and (d[0] == ncio.FILL_I2).all() and d[1, 0, 0] == ncio.FILL_I2
    back.close()
    nm = ncio.create_normals_mosaic_ds(str(tmp_path / "normals_tmax.nc"), "tmax", lon, lat, "1.3.0", format=fmt)
    assert nm.variables["tmax_normal"].dtype == np.int16 and nm.variables["tmax_se"].shape == (12, 8, 12)
    assert nm.variables["climatology_bounds"][0, 0] == (dt.date(1981, 1, 1) - dt.date(1948, 1, 1)).days
    assert nm.variables["time"].climatology == "climatology_bounds"
    nm.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_create_climdiv_optim_nstns_db_keeps_the_reference_call_shape(tmp_path, fmt):
    """optimize.py:39-82 + step21:124-128: create, then the writer rank assigns station columns into the OPEN file."""
    ids = np.array(["S0000001", "S0000002", "S0000003"])
    ladder = np.array([35, 50, 100], np.int32)
    ds = ncio.create_climdiv_optim_nstns_db(str(tmp_path), "tmin", ids, ladder, 2404, format=fmt)
    err = np.arange(36, dtype=np.float64).reshape(12, 3) - 17.0
    ds.variables["mae"][:, :, 1] = np.abs(err)
    ds.sync()
    ds.close()
    mae, nghs, back_ids = ncio.read_climdiv_optim_nstns_db(ncio.climdiv_optim_nstns_path(str(tmp_path), "tmin", 2404))
    np.testing.assert_array_equal(back_ids, ids)
    np.testing.assert_array_equal(nghs, ladder)
    np.testing.assert_array_equal(mae[:, :, 1], np.abs(err))
    assert np.isnan(mae[:, :, 0]).all() and np.isnan(mae[:, :, 2]).all()        # never written -> fill -> masked


def test_without_libhdf5_the_classic_container_takes_over(tmp_path):
    """A machine without libhdf5 (simulated: TWX_HDF5_DISABLE): writers fall back to classic netCDF with ONE warning, classic
    files read as before, and a NetCDF-4 file is refused with a message that names the library -- never a silent misread."""
    import subprocess
    import sys
    nc4 = str(tmp_path / "db4.nc")
    days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1981, 1, 5))
    grid = synth.make_grid("C1", nrows=20, ncols=20)
    db = synth.make_stations(grid["bbox"], 12, 2, "tmin", days, with_obs=True)
    if h5nc.available():
        ncio.write_station_db(nc4, db, format="NETCDF4")
    code = r"""
import sys, warnings
sys.path.insert(0, %r)
import datetime as dt, numpy as np
from topowx_amd import h5nc, ncio, synth, stationdb as sdb
from topowx_amd.dates import get_days_metadata
assert not h5nc.available()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    assert ncio.default_format() == "NETCDF3_64BIT" and ncio.default_format() == "NETCDF3_64BIT"
assert len(w) == 1 and "libhdf5" in str(w[0].message)
days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1981, 1, 5))
grid = synth.make_grid("C1", nrows=20, ncols=20)
db = synth.make_stations(grid["bbox"], 12, 2, "tmin", days, with_obs=True)
p = %r
ncio.write_station_db(p, db)                       # default container: classic now
assert ncio.file_format(p) == "NETCDF3_64BIT"
back = sdb.StationSerialDataDb(p, "tmin")
assert np.array_equal(back.var, db.var) and np.array_equal(back.stn_ids, db.stn_ids)
import os
if os.path.exists(%r):
    try:
        sdb.StationSerialDataDb(%r, "tmin")
        raise SystemExit("a NetCDF-4 file was opened without libhdf5")
    except IOError as e:
        assert "libhdf5" in str(e), e
try:
    ncio.open_dataset(p + ".x", "w", "NETCDF4")
    raise SystemExit("NETCDF4 written without libhdf5")
except IOError as e:
    assert "libhdf5" in str(e)
print("ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "db3.nc"), nc4, nc4)
    env = dict(os.environ, TWX_HDF5_DISABLE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:] + r.stdout[-500:]
