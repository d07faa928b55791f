"""Foreign-library interoperability of the NetCDF-4 containers.

netCDF4-python / netCDF-C are not in this image, but an INDEPENDENT HDF5 binding is: ``/opt/conda/bin/python3.9`` with h5py
(its own build of libhdf5).  Through a subprocess of that interpreter (tests/tools/h5py_peer.py, which imports nothing of
this repository):

* the files ``topowx_amd.ncio`` writes -- a station database, a tile written chunk by chunk through ``TileWriter``, tiles
  written by ``TileSink`` (chunks copied straight into the file's pages; chunks deflated by the sink's workers and appended
  with ``H5Dwrite_chunk``), a daily mosaic -- are opened by h5py: dimensions are dimension scales with ``_Netcdf4Dimid``,
  every variable is attached to its dimensions' scales, ``_FillValue`` / chunk shapes / filters are what the reference's
  writers ask for (tiling.py:304-537, create_db_all_stations.py:262-311), and the VALUES h5py reads through its filter
  pipeline hash to what was written;
* a station database that h5py writes the way netCDF-C's libhdf5 layer lays one out (``make_scale`` / ``attach_scale``,
  ``_Netcdf4Dimid``, variable-length string ids, gzip ``(ndays, 1)`` chunks) is read by ``StationDataWrkChk(path, 'tmin')``
  (step25:53, station_data.py:554-616) field for field.
CPU only; skipped where no interpreter with h5py exists."""
import datetime as dt
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from topowx_amd import h5nc, ncio, synth
from topowx_amd import stationdb as sdb
from topowx_amd.dates import get_days_metadata
from topowx_amd.interp import TileGridInfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEER = os.path.join(ROOT, "tests", "tools", "h5py_peer.py")


def _peer_python():
    for exe in (os.environ.get("TWX_H5PY_PYTHON"), "/opt/conda/bin/python3.9", "/opt/conda/bin/python3", "/opt/conda/bin/python"):
        if exe and os.path.exists(exe):
            env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")}
            r = subprocess.run([exe, "-c", "import h5py"], capture_output=True, env=env)
            if r.returncode == 0:
                return exe
    return None


PY = _peer_python()
pytestmark = [pytest.mark.skipif(PY is None, reason="no interpreter with h5py (looked for /opt/conda/bin/python3.9; TWX_H5PY_PYTHON)"),
              pytest.mark.skipif(not h5nc.available(), reason="libhdf5 not loadable")]


def peer(cmd, path):
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")}
    r = subprocess.run([PY, PEER, cmd, str(path)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _scales_ok(desc, dims):
    for i, d in enumerate(dims):
        dimid = desc[d]["attrs"]["_Netcdf4Dimid"]
        assert desc[d]["is_scale"] and (dimid[0] if isinstance(dimid, list) else dimid) == i, d


def test_h5py_reads_what_ncio_writes(tmp_path):
    days = get_days_metadata(dt.date(1999, 12, 25), dt.date(2000, 1, 13))
    nd = days.size
    rng = np.random.default_rng(3)
    # ---- station database
    grid = synth.make_grid("C1", nrows=20, ncols=20)
    db = synth.make_stations(grid["bbox"], 40, 9, "tmin", days, with_obs=True)
    p_db = str(tmp_path / "stn.nc")
    ncio.write_station_db(p_db, db, format="NETCDF4", zlib=True)
    d = peer("describe", p_db)
    _scales_ok(d, [sdb.STN_ID, "time"])
    assert d[sdb.STN_ID]["dtype"] == "vlen-str" and d[sdb.STN_ID]["sha256"] == hashlib.sha256("\0".join(db.stns[sdb.STN_ID]).encode()).hexdigest()
    assert d["tmin"]["dims"] == ["time", sdb.STN_ID] and d["tmin"]["chunks"] == [nd, 40] and d["tmin"]["compression"] == "gzip"
    assert d["tmin"]["shuffle"] and d["tmin"]["sha256"] == sha(np.asarray(db.var, np.float32))
    assert abs(d["tmin"]["fillvalue"][0] - 9.969209968386869e36) < 1e30                     # netCDF's default fill, no attribute
    assert "_FillValue" not in d["tmin"]["attrs"]
    assert d[sdb.LON]["dims"] == [sdb.STN_ID] and d[sdb.LON]["attrs"]["_FillValue"] == [ncio.FILL_F8]
    assert d[sdb.LON]["sha256"] == sha(np.asarray(db.stns[sdb.LON], np.float64))
    # ---- a tile through TileWriter (the reference's chunk-by-chunk writer), deflated
    lons = -110.0 + (np.arange(12) + 0.5) / 120.0
    lats = 45.0 - (np.arange(8) + 0.5) / 120.0
    info = TileGridInfo({0: "h00v00", 1: "h01v00"}, {"h00v00": (0, 0), "h01v00": (0, 6)}, 2, lons, lats, 8, 6, 4, 3, 32)
    w = ncio.TileWriter(info, str(tmp_path / "tw"), format="NETCDF4", zlib=True)
    daily = rng.integers(-3000, 3000, (nd, 8, 6)).astype(np.int16)
    norm, se, ninv = rng.normal(0, 5, (12, 8, 6)).astype("f4"), rng.random((12, 8, 6)).astype("f4"), rng.integers(0, 9, (8, 6)).astype("i4")
    for r0 in (0, 4):
        for c0 in (0, 3):
            w.write_tile_chunk("h01v00", "tmax", days, r0, c0, daily[:, r0:r0 + 4, c0:c0 + 3], norm[:, r0:r0 + 4, c0:c0 + 3],
                               se[:, r0:r0 + 4, c0:c0 + 3], ninv[r0:r0 + 4, c0:c0 + 3])
    d = peer("describe", w.fpath("h01v00", "tmax"))
    _scales_ok(d, ["time", "lat", "lon", "nv", "time_normals"])
    assert d["nv"]["attrs"]["NAME"].startswith("This is a netCDF dimension but not a netCDF variable.")
    assert d["tmax"]["dims"] == ["time", "lat", "lon"] and d["tmax"]["chunks"] == [nd, 4, 3] and d["tmax"]["compression"] == "gzip"
    assert d["tmax"]["fillvalue"] == [-32767] and d["tmax"]["attrs"]["_FillValue"] == [-32767] and d["tmax"]["sha256"] == sha(daily)
    assert abs(d["tmax"]["attrs"]["scale_factor"][0] - 0.01) < 1e-9 and d["tmax"]["attrs"]["units"] == "C"
    assert d["tmax_normal"]["dims"] == ["time_normals", "lat", "lon"] and d["tmax_normal"]["sha256"] == sha(norm)
    assert d["inconsist_tair"]["sha256"] == sha(ninv) and d["lat"]["sha256"] == sha(lats)
    assert d["/attrs"]["Conventions"] == "CF-1.6"
    # ---- tiles through TileSink: chunk bytes copied straight into the file / deflated by the workers
    arrays = {"daily_tmax": daily, "norm_tmax": norm, "se_tmax": se, "daily_tmin": (daily - 700).astype(np.int16),
              "norm_tmin": norm - 7, "se_tmin": se, "ninvalid": ninv}
    for zl in (False, True):
        sink = ncio.TileSink(info, str(tmp_path / ("sink%d" % zl)), days, threads=3, zlib=zl, verify=(0,))
        sink(0, arrays)
        sink.close()
        assert sink.stats["verified"] == 1 and sink.stats["tiles"] == 1 and sink.stats["int16_bytes"] == 2 * daily.nbytes
        for var in ("tmin", "tmax"):
            d = peer("describe", sink.writer.fpath("h00v00", var))
            _scales_ok(d, ["time", "lat", "lon", "nv", "time_normals"])
            assert d[var]["dims"] == ["time", "lat", "lon"] and d[var]["chunks"] == [nd, 4, 3], (zl, var)
            assert d[var]["compression"] == ("gzip" if zl else None) and d[var]["shuffle"] == zl
            assert d[var]["sha256"] == sha(arrays["daily_" + var]), (zl, var)                # h5py decodes what the sink stored
            assert d[var + "_normal"]["sha256"] == sha(np.asarray(arrays["norm_" + var], np.float32))
            assert d[var]["attrs"]["_FillValue"] == [-32767]
        t = ncio.read_tile(sink.writer.fpath("h00v00", "tmin"), "tmin")
        assert np.array_equal(t["daily"], arrays["daily_tmin"]) and np.array_equal(t["ninvalid"], ninv)
    # ---- chunks deflated by the GPU's encoder (here: its CPU restatement, byte-identical -- tests/test_gpu_deflate.py), appended
    #      with H5Dwrite_chunk: h5py's own HDF5 + zlib decode them
    from oracle import deflate_oracle as dorc
    pre = {k: v for k, v in arrays.items() if not k.startswith("daily_")}
    pre.update({"deflated_tmin": dorc.deflate_tile(arrays["daily_tmin"], 4, 3), "deflated_tmax": dorc.deflate_tile(arrays["daily_tmax"], 4, 3),
                "deflate_chunks": (4, 3)})
    sink = ncio.TileSink(info, str(tmp_path / "sink_gpu"), days, threads=2, zlib=True, verify=(0,))
    sink(0, pre)
    sink.close()
    for var in ("tmin", "tmax"):
        d = peer("describe", sink.writer.fpath("h00v00", var))
        assert d[var]["compression"] == "gzip" and d[var]["shuffle"] and d[var]["chunks"] == [nd, 4, 3]
        assert d[var]["sha256"] == sha(arrays["daily_" + var]), var
    # ---- a daily mosaic file
    p_m = str(tmp_path / "mosaic.nc")
    ds = ncio.create_dly_mosaic_ds(p_m, "tmin", days, lons, lats, "1.2.3", format="NETCDF4")
    ds.variables["tmin"][:] = daily.repeat(2, axis=2)
    ds.close()
    d = peer("describe", p_m)
    assert d["tmin"]["dims"] == ["time", "lat", "lon"] and d["tmin"]["chunks"] == [1, 8, 12] and d["tmin"]["compression"] == "gzip"
    assert d["tmin"]["sha256"] == sha(daily.repeat(2, axis=2)) and d["time_bnds"]["dims"] == ["time", "nv"]


def test_ncio_reads_what_h5py_writes_the_netcdf_c_way(tmp_path):
    p = str(tmp_path / "peer_db.nc")
    wrote = peer("write-station-db", p)
    assert ncio.file_format(p) == "NETCDF4"
    da = sdb.StationDataWrkChk(p, "tmin")                                    # step25:53
    np.testing.assert_array_equal(da.stn_ids, wrote["ids"])
    assert da.days.size == wrote["ndays"] and da.days.YMD[0] == wrote["first_day"]
    np.testing.assert_array_equal(da.stns[sdb.LON], wrote["lon"])
    np.testing.assert_array_equal(da.stns[sdb.LAT], wrote["lat"])
    np.testing.assert_array_equal(da.stns[sdb.ELEV], wrote["elev"])
    n01 = np.array([np.nan if v is None else v for v in wrote["norm01"]])
    np.testing.assert_array_equal(da.stns["norm01"], n01)                    # default fill without attribute -> NaN
    assert da.stns["station_name"][5] == "STATION 05" and da.stns[sdb.TDI].shape == (len(wrote["ids"]),)
    assert sha(np.asarray(da.var, np.float32)) == wrote["obs_sha256"]
    ds = h5nc.Dataset(p)
    v = ds.variables["tmin"]
    assert v.dimensions == ("time", sdb.STN_ID) and v.chunking() == [wrote["ndays"], 1]
    assert v.filters() == {"zlib": True, "shuffle": True, "complevel": 4}
    assert list(ds.dimensions) == ["time", sdb.STN_ID] and ds.variables[sdb.STN_ID].dimensions == (sdb.STN_ID,)
    ds.close()
