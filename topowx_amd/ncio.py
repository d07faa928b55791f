"""netCDF containers either side of the path.

The reference reads its serially-complete station DB and writes its tiles, mosaics and cross-validation files through
netCDF4-python (``twx/db/station_data.py:547-666``, ``twx/interp/tiling.py:304-537,553-1078,1169-1219``,
``twx/interp/optimize.py:39-82``).  Two containers are supported here behind ONE dataset interface (netCDF4-python's
names: ``dimensions``, ``variables``, ``createDimension``, ``createVariable``, attribute syntax, hyperslab indexing):

* **NETCDF4** -- the reference's own format -- through ``topowx_amd.h5nc``: ``ctypes`` on the HDF5 library of the image
  (dimension scales, chunking, zlib + shuffle, variable-length string ids).  The default when libhdf5 can be loaded.
* **NETCDF3_64BIT** (classic, 64-bit offset) through ``scipy.io.netcdf_file``: no chunking / compression, station ids as
  ``char[station_id][string<N>]`` arrays (the layout ``_build_stn_struct`` also accepts, station_data.py:134-157).  The
  fallback when libhdf5 is absent (one warning), or on request (``format="NETCDF3_64BIT"``).

Readers pick the container from the file's magic bytes; every writer takes ``format=``.
"""
import datetime as _dt
import os
import threading
import warnings

import numpy as np
from scipy.io import netcdf_file

from . import h5nc
from . import stationdb as sdb
from .dates import DAY, MONTH, YEAR, get_days_metadata

__all__ = ["TileWriter", "read_tile", "write_station_db", "read_station_db", "read_station_db_arrays", "read_tile_stores",
           "CONVERT_HELP", "climdiv_optim_nstns_path", "create_climdiv_optim_nstns_db", "write_climdiv_optim_nstns_db",
           "read_climdiv_optim_nstns_db", "create_quick_db", "open_dataset", "default_format", "file_format", "convert_station_db", "FORMATS",
           "create_dly_mosaic_ds", "create_normals_mosaic_ds", "create_ds_mthly", "TWX_SOURCE"]

FILL_I2 = np.int16(-32767)
FILL_F4 = np.float32(9.969209968386869e36)
FILL_I4 = np.int32(-2147483647)
FILL_F8 = 9.969209968386869e36          # netCDF4.default_fillvals['f8'] (create_db_all_stations.py:164)
# netCDF4.default_fillvals: what netCDF4-python's auto-mask hides in a variable WITHOUT a _FillValue attribute
# (never-written entries of a variable created with fill_value=None read back masked -> NaN, station_data.py:159-164)
DEFAULT_FILLS = {"f4": float(FILL_F4), "f8": FILL_F8, "i2": -32767, "i4": -2147483647, "i8": -9223372036854775806,
                 "u2": 65535, "u4": 4294967295}
SCALE_FACTOR = np.float32(0.01)                      # tiling.py:36
# long name, units, standard name, cell method (tiling.py:39-42)
VAR_ATTRS = {"tmin": ("minimum air temperature", "C", "air_temperature", "minimum"),
             "tmax": ("maximum air temperature", "C", "air_temperature", "maximum")}
FORMATS = ("NETCDF4", "NETCDF3_64BIT")
TWX_SOURCE = "topowx_amd (MI355X implementation of the TopoWx interpolation path)"
_warned = []


def default_format():
    """NETCDF4 (the reference's container) when the HDF5 library is loadable, classic netCDF otherwise."""
    if h5nc.available():
        return "NETCDF4"
    if not _warned:
        _warned.append(1)
        warnings.warn("libhdf5 / libhdf5_hl not found: topowx_amd.ncio writes classic netCDF (NETCDF3_64BIT) instead of "
                      "NetCDF-4 and cannot open NetCDF-4 / HDF5 files (set TWX_HDF5_LIBDIR to the library's directory)")
    return "NETCDF3_64BIT"


def file_format(path):
    """'NETCDF4' / 'NETCDF3_64BIT' / 'NETCDF3_CLASSIC' from the magic bytes; IOError for anything else."""
    if not os.path.exists(path):
        raise IOError("No such file or directory: %s" % path)
    with open(path, "rb") as fh:
        magic = fh.read(8)
    if magic[:3] == b"CDF":
        return "NETCDF3_64BIT" if magic[3:4] == b"\x02" else "NETCDF3_CLASSIC"
    if magic == b"\x89HDF\r\n\x1a\n":
        return "NETCDF4"
    raise IOError("%s is neither a classic netCDF nor a NetCDF-4 / HDF5 file (magic %r)" % (path, magic[:4]))


# ---- classic netCDF behind the NetCDF-4 interface ---------------------------------------------------------------------
_CODES = {"f8": "d", "f4": "f", "i4": "i", "i2": "h", "i1": "b", "S1": "c"}


class _ClassicVar(object):
    _OWN = ("_v", "_ds", "name")

    def __init__(self, ds, name, v):
        object.__setattr__(self, "_v", v)
        object.__setattr__(self, "_ds", ds)
        object.__setattr__(self, "name", name)

    dimensions = property(lambda self: tuple(self._v.dimensions))
    shape = property(lambda self: tuple(self._v.shape))
    ndim = property(lambda self: len(self._v.shape))

    @property
    def dtype(self):
        dt = np.dtype(self._v.data.dtype) if hasattr(self._v, "data") else np.dtype(self._v.typecode())
        return dt.newbyteorder("=") if dt.kind != "S" else dt

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        a = np.asarray(self._v[key] if self.shape else self._v.getValue())
        if a.dtype.kind != "S":
            a = a.astype(a.dtype.newbyteorder("="))         # classic data is big-endian
        return a.copy() if a.ndim else a[()]

    def __setitem__(self, key, value):
        if np.ma.isMaskedArray(value):
            fill = self._v._attributes.get("_FillValue")
            value = np.ma.filled(value, fill if fill is not None else DEFAULT_FILLS.get(self.dtype.str[1:], 0))
        if self.shape:
            self._v[key] = value
        else:
            self._v.assignValue(value)

    def ncattrs(self):
        return list(self._v._attributes)

    def getncattr(self, name):
        if name not in self._v._attributes:
            raise AttributeError("NetCDF: Attribute not found: %s" % name)
        return _attr_out(self._v._attributes[name])

    def setncattr(self, name, value):
        setattr(self._v, name, _attr_in(value))

    def chunking(self):
        return "contiguous"

    def filters(self):
        return {"zlib": False, "shuffle": False, "complevel": 0}

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self.getncattr(name)

    def __setattr__(self, name, value):
        if name in self._OWN:
            object.__setattr__(self, name, value)
        else:
            self.setncattr(name, value)


def _attr_out(v):
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    if isinstance(v, str):
        return str(v)
    a = np.asarray(v)
    if a.dtype.kind == "S":
        return a.tobytes().decode("utf-8", "replace")
    a = a.astype(a.dtype.newbyteorder("="))
    return a[()] if a.ndim == 0 else (a[0] if a.size == 1 else a)


def _attr_in(v):
    if isinstance(v, (str, bytes)):
        return str(v) if isinstance(v, str) else v
    a = np.asarray(v)
    if a.dtype.kind == "U":
        return " ".join(str(x) for x in np.atleast_1d(a).tolist())
    if a.dtype == np.int64:                              # classic netCDF has no 64-bit integers
        a = a.astype(np.int32)
    return a


class _Classic(object):
    """``scipy.io.netcdf_file`` with netCDF4-python's method names (a variable's data stays in memory until close)."""
    _OWN = ("_nc", "path", "mode", "variables", "data_model")

    def __init__(self, path, mode="r", version=2):
        object.__setattr__(self, "path", os.fspath(path))
        object.__setattr__(self, "mode", mode)
        object.__setattr__(self, "data_model", "NETCDF3_64BIT" if version == 2 else "NETCDF3_CLASSIC")
        if mode in ("a", "r+"):
            nc = netcdf_file(self.path, "a", mmap=False)
        elif mode == "w":
            nc = netcdf_file(self.path, "w", version=version, mmap=False)
        else:
            nc = netcdf_file(self.path, "r", mmap=False)
        object.__setattr__(self, "_nc", nc)
        object.__setattr__(self, "variables", {k: _ClassicVar(self, k, v) for k, v in nc.variables.items()})

    @property
    def dimensions(self):
        return dict(self._nc.dimensions)

    def createDimension(self, name, size):
        self._nc.createDimension(name, None if size is None else int(size))

    def createVariable(self, varname, datatype, dimensions=(), zlib=False, complevel=4, shuffle=True, chunksizes=None,
                       fill_value=None, contiguous=False):
        if datatype is str or datatype == "str":
            raise TypeError("classic netCDF has no string type: use a char array (station_id, string<N>)")
        dt = np.dtype(datatype)
        key = "S1" if dt.kind == "S" else dt.str[1:]
        if key not in _CODES:
            raise TypeError("classic netCDF has no type for %r" % (dt,))
        if isinstance(dimensions, str):
            dimensions = (dimensions,)
        v = self._nc.createVariable(varname, _CODES[key], tuple(dimensions))
        var = _ClassicVar(self, varname, v)
        self.variables[varname] = var
        if fill_value is not None and fill_value is not False:
            fv = np.array(fill_value).astype(dt)[()]
            v._FillValue = fv
            if v.shape:
                v[:] = fv                                   # the classic format has no implicit fill on this writer
        return var

    def ncattrs(self):
        return list(self._nc._attributes)

    def getncattr(self, name):
        if name not in self._nc._attributes:
            raise AttributeError("NetCDF: Attribute not found: %s" % name)
        return _attr_out(self._nc._attributes[name])

    def setncattr(self, name, value):
        setattr(self._nc, name, _attr_in(value))

    def sync(self):
        if self.mode != "r":
            self._nc.flush()

    def close(self):
        if self._nc is not None:
            self._nc.close()
            object.__setattr__(self, "_nc", None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self.getncattr(name)

    def __setattr__(self, name, value):
        if name in self._OWN:
            object.__setattr__(self, name, value)
        else:
            self.setncattr(name, value)


def open_dataset(path, mode="r", format=None, rdcc_nbytes=None, alignment=None):
    """``netCDF4.Dataset(path, mode)`` on either container.  'w': ``format`` or ``default_format()``; otherwise the file's
    own format (magic bytes).  ``alignment``: see ``h5nc.Dataset`` (NETCDF4 only)."""
    path = os.fspath(path)
    if mode == "w":
        fmt = format or default_format()
        if fmt == "NETCDF4":
            return h5nc.Dataset(path, "w", alignment=alignment)
        if fmt in ("NETCDF3_64BIT", "NETCDF3_CLASSIC"):
            return _Classic(path, "w", version=2 if fmt == "NETCDF3_64BIT" else 1)
        raise ValueError("format must be one of %r" % (FORMATS,))
    fmt = file_format(path)
    if fmt == "NETCDF4":
        if not h5nc.available():
            raise IOError("%s is a NetCDF-4 / HDF5 file and libhdf5 could not be loaded (TWX_HDF5_LIBDIR); see "
                          "python -m topowx_amd.ncio --convert-help" % path)
        return h5nc.Dataset(path, mode, rdcc_nbytes=rdcc_nbytes)
    return _Classic(path, mode)


def _is_nc4(ds):
    return getattr(ds, "data_model", "") == "NETCDF4"


def _date(days, i):
    return _dt.date(int(days[YEAR][i]), int(days[MONTH][i]), int(days[DAY][i]))


def _num(d, d0):
    return float((d - d0).days)


def _mid(a, b):
    return a + _dt.timedelta(days=((b - a).days // 2))


def _ymd(d):
    return d.year * 10000 + d.month * 100 + d.day


def _units(d0):
    return "days since %d-%d-%d 0:0:0" % (d0.year, d0.month, d0.day)


def _parse_units(units):
    if not units.startswith("days since "):
        raise ValueError("time units must be 'days since ...'")
    y, m, d = (int(x) for x in units.split()[2].split("-"))
    return _dt.date(y, m, d)


def days_of(ds):
    """``get_days_metadata_dates(num2date(time[:], time.units))`` for a daily, gap-free axis (station_data.py:574-576,
    tiling.py:616-627)."""
    tv = ds.variables["time"]
    d0 = _parse_units(tv.units)
    t = np.floor(np.asarray(tv[:], np.float64)).astype(np.int64)
    if not np.array_equal(t - t[0], np.arange(t.size)):
        raise ValueError("time axis must be daily and gap-free")
    return get_days_metadata(d0 + _dt.timedelta(days=int(t[0])), d0 + _dt.timedelta(days=int(t[-1])))


def _time_vars(ds, days, units, bounds=True):
    """``time`` (+ ``time_bnds``): day midpoints relative to ``units`` (tiling.py:378-397,697-716)."""
    d0 = _parse_units(units)
    times = ds.createVariable("time", "f8", ("time",))
    times.long_name, times.units, times.standard_name, times.calendar = "time", units, "time", "standard"
    first = _num(_date(days, 0), d0)
    nums = first + np.arange(days.size, dtype=np.float64) + 0.5
    times[:] = nums
    if bounds:
        times.bounds = "time_bnds"
        tb = ds.createVariable("time_bnds", "f8", ("time", "nv"))
        tb[:] = np.stack([nums - 0.5, nums + 0.5], axis=1)
    return times


def _lonlat_vars(ds, lons, lats):
    la = ds.createVariable("lat", "f8", ("lat",))
    la.long_name, la.units, la.standard_name = "latitude", "degrees_north", "latitude"
    la[:] = lats
    lo = ds.createVariable("lon", "f8", ("lon",))
    lo.long_name, lo.units, lo.standard_name = "longitude", "degrees_east", "longitude"
    lo[:] = lons


def _crs_var(ds):
    """``_add_crs_wgs84_var`` (tiling.py:539-546)."""
    crs = ds.createVariable("crs", "i2", ())
    crs.grid_mapping_name = "latitude_longitude"
    crs.longitude_of_prime_meridian = 0.0
    crs.semi_major_axis = 6378137.0
    crs.inverse_flattening = 298.257223563


def _gridded(v):
    v.coordinates, v.grid_mapping = "lat lon", "crs"


def _climatology(ds, dim, d0):
    """The 1981-2010 normals' time axis and its bounds (tiling.py:399-420,872-895)."""
    tn = ds.variables[dim]
    cb = ds.createVariable("climatology_bounds", "f8", (dim, "nv"))
    for m in range(1, 13):
        mn, y1 = (m + 1, 1981) if m != 12 else (1, 1982)
        tn[m - 1] = _num(_mid(_dt.date(1981, m, 1), _dt.date(y1, mn, 1)), d0)
        cb[m - 1, :] = [_num(_dt.date(1981, m, 1), d0), _num(_dt.date(2010 if m != 12 else 2011, mn, 1), d0)]


class TileWriter(object):
    """``TileWriter`` (tiling.py:304-537): one file ``<path_out>/<tile_id>/<tile_id>_<varname>.nc`` per tile
    and variable, created on the first chunk, reopened for every further chunk.  NETCDF4 files carry the reference's
    chunk shapes (``(ndays, chk_size_y, chk_size_x)`` for the daily variable, tiling.py:453-455); ``zlib`` is off as in
    the reference's tiles."""

    def __init__(self, tile_grid_info, path_out, format=None, zlib=False, complevel=4):
        t = tile_grid_info
        self.tile_ids, self.tile_rc, self.ntiles = t.tile_ids, t.tile_rc, t.ntiles
        self.lons, self.lats = np.asarray(t.lons, np.float64), np.asarray(t.lats, np.float64)
        self.path_out = path_out
        self.tile_size_y, self.tile_size_x = t.tile_size_y, t.tile_size_x
        self.chk_size_y, self.chk_size_x = t.chk_size_y, t.chk_size_x
        self.format, self.zlib, self.complevel = format, zlib, complevel

    def fpath(self, tile_id, varname):
        return os.path.join(self.path_out, tile_id, "%s_%s.nc" % (tile_id, varname))

    def _create(self, fpath, tile_id, varname, days, early=False):
        """``early`` (NETCDF4, ``TileSink``): the daily variable's chunks are allocated at creation, on page boundaries, and
        not pre-filled -- the caller writes every one of them."""
        os.makedirs(os.path.dirname(fpath), exist_ok=True)
        ds = open_dataset(fpath, "w", self.format, alignment=(1 << 20, 4096) if early else None)
        d0, d1 = _date(days, 0), _date(days, days.size - 1)
        ds.title = "Daily Interpolated Meteorological Data %d-%d" % (_ymd(d0), _ymd(d1))
        ds.institution = "University of Montana"
        ds.source = TWX_SOURCE
        ds.history = "Created on: " + _dt.date.today().strftime("%Y-%m-%d")
        ds.references = "http://www.ntsg.umt.edu/project/TopoWx"
        ds.comment = "30-arcsec spatial resolution, daily timestep"
        ds.Conventions = "CF-1.6"

        str_row, str_col = self.tile_rc[tile_id]
        lons = self.lons[str_col:str_col + self.tile_size_x]
        lats = self.lats[str_row:str_row + self.tile_size_y]
        ds.createDimension("time", int(days.size))
        ds.createDimension("lat", int(lats.size))
        ds.createDimension("lon", int(lons.size))
        ds.createDimension("nv", 2)
        ds.createDimension("time_normals", 12)

        units = _units(d0)
        _time_vars(ds, days, units)
        tn = ds.createVariable("time_normals", "f8", ("time_normals",))
        tn.long_name, tn.units, tn.standard_name, tn.calendar = "time", units, "time", "standard"
        tn.climatology = "climatology_bounds"
        tn.comment = "Time dimension for the 1981-2010 monthly normals"
        _climatology(ds, "time_normals", d0)
        _lonlat_vars(ds, lons, lats)
        _crs_var(ds)

        long_name, vunits, std_name, cell_method = VAR_ATTRS[varname]
        cy, cx = min(self.chk_size_y, lats.size), min(self.chk_size_x, lons.size)
        kw = dict(zlib=self.zlib, complevel=self.complevel)
        if early and _is_nc4(ds) and not self.zlib:
            kw["alloc_early"] = True                          # (TileSink: the file has its final size once it is created)
        mv = ds.createVariable(varname, "i2", ("time", "lat", "lon"), chunksizes=(int(days.size), cy, cx),
                               fill_value=FILL_I2, **kw)
        mv.long_name, mv.units, mv.standard_name = long_name, vunits, std_name
        mv.scale_factor = SCALE_FACTOR
        mv.cell_methods = "area: mean time: " + cell_method
        _gridded(mv)
        nv = ds.createVariable(varname + "_normal", "f4", ("time_normals", "lat", "lon"), chunksizes=(12, cy, cx),
                               fill_value=FILL_F4, **kw)
        nv.long_name, nv.units, nv.standard_name = "normal " + long_name, vunits, std_name
        nv.ancillary_variables = varname + "_se"
        nv.comment = "The 1981-2010 monthly normals"
        nv.cell_methods = "time: %s within years time: mean over years" % cell_method
        _gridded(nv)
        sv = ds.createVariable(varname + "_se", "f4", ("time_normals", "lat", "lon"), chunksizes=(12, cy, cx),
                               fill_value=FILL_F4, **kw)
        sv.long_name = long_name + " kriging standard error"
        sv.standard_name, sv.units = "air_temperature standard_error", vunits
        sv.comment = "The uncertainty in the 1981-2010 monthly normals"
        _gridded(sv)
        iv = ds.createVariable("inconsist_tair", "i4", ("lat", "lon"), chunksizes=(cy, cx), fill_value=FILL_I4, **kw)
        iv.long_name, iv.units = "number of days interpolated tmin >= tmax", "days"
        iv.comment = ("The number of days daily tmin/tmax had to be adjusted due to interpolated tmin "
                      "being >= interpolated tmax")
        _gridded(iv)
        return ds

    def write_tile_chunk(self, tile_id, varname, days, str_row, str_col, daily_vals, mthly_normals,
                         mthly_normals_se, ninvalid):
        """tiling.py:488-537; ``daily_vals`` is the packed int16 block (or None for a normals-only run)."""
        fpath = self.fpath(tile_id, varname)
        ds = open_dataset(fpath, "a") if os.path.exists(fpath) else self._create(fpath, tile_id, varname, days)
        try:
            ny, nx = np.asarray(mthly_normals).shape[-2:]
            rs, cs = slice(str_row, str_row + ny), slice(str_col, str_col + nx)
            if daily_vals is not None:
                ds.variables[varname][:, rs, cs] = np.asarray(daily_vals, np.int16)
            ds.variables[varname + "_normal"][:, rs, cs] = np.asarray(mthly_normals, np.float32)
            ds.variables[varname + "_se"][:, rs, cs] = np.asarray(mthly_normals_se, np.float32)
            ds.variables["inconsist_tair"][rs, cs] = np.asarray(ninvalid, np.int32)
        finally:
            ds.close()


def read_tile(fpath, varname):
    """Arrays of one tile file (either container): daily int16 (raw), normals / SE f4, ninvalid, lon, lat, time."""
    ds = open_dataset(fpath, "r")
    try:
        g = ds.variables
        out = {"daily": g[varname][:], "norm": g[varname + "_normal"][:], "se": g[varname + "_se"][:],
               "ninvalid": g["inconsist_tair"][:], "lon": g["lon"][:], "lat": g["lat"][:], "time": g["time"][:],
               "time_units": g["time"].units, "scale_factor": g[varname].scale_factor}
    finally:
        ds.close()
    return out


def read_tile_stores(path_in, tiles, variables=("tmin", "tmax")):
    """Tile files -> ``{tile_id: TileStore}`` for ``TileMosaic`` (missing tiles are skipped, as the
    mosaicker treats them: tiling.py:772-776)."""
    from .step25 import TileStore
    stores = {}
    for t in tiles:
        st = None
        for v in variables:
            fp = os.path.join(path_in, t, "%s_%s.nc" % (t, v))
            if not os.path.exists(fp):
                continue
            a = read_tile(fp, v)
            if st is None:
                st = TileStore(a["daily"].shape[0], a["norm"].shape[1], a["norm"].shape[2], True)
            st.a["daily_" + v], st.a["norm_" + v], st.a["se_" + v] = a["daily"], a["norm"], a["se"]
            st.a["ninvalid"] = a["ninvalid"]
        if st is not None:
            stores[t] = st
    return stores


class TileSink(object):
    """``sink(tile_number, arrays)`` of ``driver.interp_tiles_streamed``: every finished tile goes into the reference's
    per-tile NetCDF-4 files ``<path_out>/<tile_id>/<tile_id>_<var>.nc`` (``TileWriter``'s layout: tiling.py:304-537) at the
    rate the host's memory system takes them, not at the rate of one ``H5Dwrite`` thread.

    The reference's worker re-opens the tile's file for every 50 x 50 chunk and writes its ``(ndays, 50, 50)`` int16 block
    through netCDF4-python (step25:177-185, tiling.py:488-537); a 250 x 250 tile of configs[3] is 25 such chunks of 126 MB
    per variable, and one GPU produces a tile every ~0.12 s.  Here, per tile and variable:

    * PREPARE (needs no data; with ``order`` -- the tile numbers in the order they will arrive -- the files of the next ``ahead``
      tiles are prepared by background threads while this tile is written): the file is created with EVERY variable's chunks
      allocated at creation (``h5nc``: ``alloc_early``), coordinates and attributes are written, the chunk addresses are read
      (``H5Dget_chunk_info``), the file is closed, and its pages are allocated with ONE ``posix_fallocate`` (17-19 GB/s per
      file on the GPU box; new file pages by first touch: 1.3-4 GB/s per file however many threads);
    * WRITE: ``threads`` workers gather the tile's ``[ndays, Y, X]`` block (the pinned host slot the GPU's outputs arrived in)
      into a warm, reused STAGING buffer in the file's chunk order -- strided gathers of (days segment x chunk) pieces, numpy
      releases the GIL: 50-65 GB/s with 32 threads --, and one thread per FILE ``pwrite``s every chunk to its address as soon
      as it is complete (11 GB/s per file into allocated pages, and files add up: 21 GB/s for a tile's two, 30 for four).  No
      HDF5 call touches the 6.3 GB of a tile, no page is faulted in or mapped; the small variables (normals, SE,
      inconsist_tair) go through the library into their allocated chunks meanwhile.
      (Round 6 went through five forms that copied through a shared ``mmap`` of the file instead: they end at ~10 GB/s, what
      this host gives a process that maps freshly allocated tmpfs pages whatever the number of threads, files or tiles in
      flight -- EXPERIMENTS.md.)
    * ``zlib=True`` (the reference's tiles are not compressed; its mosaics are): the workers gather, byte-shuffle and
      deflate each chunk (``zlib`` releases the GIL) and the sink's own thread appends the stored bytes with
      ``H5Dwrite_chunk`` -- for the library a byte copy.  A tile that arrives with ``deflated_<var>`` lists (a stream with
      ``deflate_chunks``: the GPU formed the chunk bytes, twx_stream_deflate) is appended as it is: no core deflates.  The files
      are prepared ahead like the plain ones, extended by ``deflate_prealloc`` x the int16 size so that the appends go into
      allocated pages (the library truncates a file to its end-of-allocation when it closes it).

    Thread-safe: with ``driver.interp_tiles_streamed(writer_threads=2)`` two tiles (four files) are written at once.
    The files are complete NetCDF-4 files of the ``TileWriter`` layout (``read_tile`` / ``h5nc.Dataset`` / any HDF5 reader
    open them; tests/test_ncio.py, tests/test_h5py_interop.py).  ``stats``: tiles, int16 bytes handed over, bytes on disk,
    seconds waiting for prepared files / in the gather + write / total (summed over concurrent calls).  ``verify``: tile
    numbers to read back through the library and compare with what was handed over (``stats["verified"]``: number of tiles
    checked, raises on a difference)."""

    def __init__(self, tile_grid_info, path_out, days, threads=None, zlib=False, complevel=1, verify=(), day_segments=None,
                 variables=("tmin", "tmax"), order=None, ahead=2, prep_threads=2, deflate_prealloc=0.75):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        if not h5nc.available():
            raise IOError("TileSink needs libhdf5 (NetCDF-4 tiles); TileWriter writes classic netCDF without it")
        self.info, self.path_out, self.days, self.variables = tile_grid_info, path_out, days, tuple(variables)
        self.writer = TileWriter(tile_grid_info, path_out, format="NETCDF4", zlib=zlib, complevel=complevel)
        self.threads = int(threads or min(32, os.cpu_count() or 8))
        self.zlib, self.complevel, self.verify = bool(zlib), int(complevel), set(verify)
        self.deflate_prealloc = float(deflate_prealloc)         # zlib: pages allocated ahead, as a fraction of the int16 size
        self.pool = ThreadPoolExecutor(self.threads)             # gathers / deflates
        self.write_pool = ThreadPoolExecutor(8)                  # one pwrite loop per file in flight
        self._lock = threading.Lock()                           # several tiles may be written at once (driver: writer_threads)
        self._staging = queue.LifoQueue()                       # warm buffers of one variable's chunk region, reused
        self.ahead = max(1, int(ahead))                         # tiles prepared ahead of the one being written (6.3 GB of pages each)
        self.prep_pool = ThreadPoolExecutor(max(1, int(prep_threads)))
        self.order = list(order) if order is not None else None
        self._pos = {k: i for i, k in enumerate(self.order)} if self.order is not None else {}
        self._ahead = {}                                        # tile number -> {var: future of _prepare_var}
        self._taken = set()                                     # tile numbers whose files have been handed to a writer
        cy, cx = tile_grid_info.chk_size_y, tile_grid_info.chk_size_x
        nchunks = len(self.variables) * (tile_grid_info.tile_size_y // cy) * (tile_grid_info.tile_size_x // cx)
        # enough gather jobs per tile to keep every worker busy: chunks are cut along the day axis
        self.day_segments = int(day_segments or max(1, -(-2 * self.threads // max(nchunks, 1))))
        self.stats = {"tiles": 0, "int16_bytes": 0, "disk_bytes": 0, "prepare_s": 0.0, "copy_s": 0.0, "total_s": 0.0, "fallocate_s": 0.0,
                      "verified": 0, "threads": self.threads, "zlib": self.zlib, "complevel": self.complevel if zlib else None,
                      "look_ahead": self.order is not None}

    # ---- prepare: no data needed ----------------------------------------------------------------------------------------------
    def _prepare_var(self, k, var):
        import time as _t
        tile_id = self.info.get_tile_id(k)
        fpath = self.writer.fpath(tile_id, var)
        ds = self.writer._create(fpath, tile_id, var, self.days, early=True)
        if self.zlib:
            # deflated chunks have no address before they exist: the library appends them (H5Dwrite_chunk).  But the pages it will
            # append into can exist: the file is extended to what the chunks will about need (libhdf5 opens a file longer than
            # its end-of-allocation and truncates it to that when it closes), so the appends are copies into allocated pages
            ds.close()
            t0 = _t.perf_counter()
            raw = 2 * self.days.size * self.info.tile_size_y * self.info.tile_size_x
            fd = os.open(fpath, os.O_RDWR)
            try:
                os.posix_fallocate(fd, 0, os.fstat(fd).st_size + int(self.deflate_prealloc * raw))
            except OSError:
                pass
            finally:
                os.close(fd)
            return fpath, None, None, _t.perf_counter() - t0
        ds.sync()
        info = ds.variables[var].chunk_info()
        ds.close()
        t0 = _t.perf_counter()
        fd = os.open(fpath, os.O_RDWR)
        try:
            os.posix_fallocate(fd, 0, os.fstat(fd).st_size)   # every page of the file, in one in-kernel loop (releases the GIL)
        except OSError:
            pass                                                # (a file system without fallocate: pwrite allocates as it goes)
        return fpath, fd, info, _t.perf_counter() - t0

    def _prepared(self, k):
        with self._lock:
            self._taken.add(k)
            futs = self._ahead.pop(k, None)
            if futs is None:
                futs = {v: self.prep_pool.submit(self._prepare_var, k, v) for v in self.variables}
            if k in self._pos:                                  # the next tiles' files, while this one is written
                for i in range(self._pos[k] + 1, min(self._pos[k] + 1 + self.ahead, len(self.order))):
                    nk = self.order[i]
                    if nk not in self._ahead and nk not in self._taken:
                        self._ahead[nk] = {v: self.prep_pool.submit(self._prepare_var, nk, v) for v in self.variables}
        return {v: f.result() for v, f in futs.items()}

    def _add(self, **kw):
        with self._lock:
            for name, v in kw.items():
                self.stats[name] += v

    def _stage(self, nbytes):
        import queue
        try:
            buf = self._staging.get_nowait()
            if buf.size >= nbytes:
                return buf
        except queue.Empty:
            pass
        return np.empty(nbytes, np.uint8)

    @staticmethod
    def _gather(dst, src, d0, d1, r0, c0):
        ny, nx = min(dst.shape[1], src.shape[1] - r0), min(dst.shape[2], src.shape[2] - c0)
        if (ny, nx) != dst.shape[1:]:
            dst[d0:d1] = FILL_I2                                # an edge chunk is stored whole: fill value beyond the tile
        np.copyto(dst[d0:d1, :ny, :nx], src[d0:d1, r0:r0 + ny, c0:c0 + nx])

    @staticmethod
    def _write_file(fd, chunks):
        """``chunks``: (file address, staging view, gather futures) in file order -- each chunk goes out as soon as it is whole."""
        try:
            for addr, view, futs in chunks:
                for f in futs:
                    f.result()
                mv = memoryview(view)
                done = 0
                while done < len(mv):                           # (pwrite may write less than asked: 2 GiB per call at most)
                    done += os.pwrite(fd, mv[done:], addr + done)
        finally:
            os.close(fd)

    @staticmethod
    def _deflate(src, r0, c0, cy, cx, level):
        import zlib as _z
        blk = np.ascontiguousarray(src[:, r0:r0 + cy, c0:c0 + cx])
        if blk.shape[1:] != (cy, cx):                          # an edge chunk is stored whole: pad with the fill value
            full = np.full((src.shape[0], cy, cx), FILL_I2, np.int16)
            full[:, :blk.shape[1], :blk.shape[2]] = blk
            blk = full
        shuf = np.ascontiguousarray(blk.reshape(-1).view(np.uint8).reshape(-1, 2).T)      # HDF5's shuffle: byte planes
        return _z.compress(shuf, level)

    @staticmethod
    def _inflate_tile(blobs, shape, cy, cx):
        """Chunk streams (row-major chunk order) -> the int16 tile they hold: inflate, undo HDF5's shuffle."""
        import zlib as _z
        out = np.empty(shape, np.int16)
        it = iter(blobs)
        for r0 in range(0, shape[1], cy):
            for c0 in range(0, shape[2], cx):
                raw = np.frombuffer(_z.decompress(next(it)), np.uint8)
                n = raw.size // 2
                out[:, r0:r0 + cy, c0:c0 + cx] = np.stack([raw[:n], raw[n:]], axis=1).reshape(-1).view("<i2").reshape(shape[0], cy, cx)
        return out

    def _small(self, ds, var, arrays):
        g = ds.variables
        g[var + "_normal"][:] = np.asarray(arrays["norm_" + var], np.float32)
        g[var + "_se"][:] = np.asarray(arrays["se_" + var], np.float32)
        g["inconsist_tair"][:] = np.asarray(arrays["ninvalid"], np.int32)

    def __call__(self, k, arrays):
        import time as _t
        t_begin = _t.perf_counter()
        tile_id = self.info.get_tile_id(k)
        cy, cx = self.info.chk_size_y, self.info.chk_size_x
        have = [v for v in self.variables if arrays.get("norm_" + v) is not None]
        # (a normals-only run: the small variables alone; a stream with deflate_chunks: the chunk bytes as the GPU formed them)
        present = [v for v in have if arrays.get("daily_" + v) is not None or arrays.get("deflated_" + v) is not None]
        if any(arrays.get("deflated_" + v) is not None for v in present):
            if not self.zlib or tuple(arrays["deflate_chunks"]) != (cy, cx):
                raise IOError("deflated chunks need a TileSink with zlib=True and the stream's chunk shape %r" % ((cy, cx),))
        jobs, open_ds, writes, stages = [], [], [], []
        if self.zlib:
            prepared = self._prepared(k)
            for var in self.variables:
                if var not in have:
                    os.remove(prepared[var][0])
            for var in have:
                self._add(fallocate_s=prepared[var][3])
                ds = open_dataset(prepared[var][0], "a")
                self._small(ds, var, arrays)
                open_ds.append(ds)
                if var not in present:
                    continue
                if arrays.get("deflated_" + var) is not None:  # already shuffled + deflated, chunks in row-major order
                    offs = [(0, r0, c0) for r0 in range(0, self.info.tile_size_y, cy) for c0 in range(0, self.info.tile_size_x, cx)]
                    jobs += [(None, ds.variables[var], off, blob) for off, blob in zip(offs, arrays["deflated_" + var])]
                    continue
                src = arrays["daily_" + var]
                for r0 in range(0, src.shape[1], cy):
                    for c0 in range(0, src.shape[2], cx):
                        jobs.append((self.pool.submit(self._deflate, src, r0, c0, cy, cx, self.complevel), ds.variables[var], (0, r0, c0), None))
            self._add(prepare_s=_t.perf_counter() - t_begin)
            t1 = _t.perf_counter()
            for fut, var_obj, off, blob in jobs:
                var_obj.write_chunk_raw(off, blob if fut is None else fut.result())      # (raises what a worker raised)
            for ds in open_ds:
                ds.close()
        else:
            prepared = self._prepared(k)
            self._add(prepare_s=_t.perf_counter() - t_begin)
            t1 = _t.perf_counter()
            for var in present:
                src = arrays["daily_" + var]
                fpath, fd, info, dt_f = prepared[var]
                self._add(fallocate_s=dt_f)
                nd = src.shape[0]
                size = nd * cy * cx * 2
                stg = self._stage(size * len(info))
                stages.append(stg)
                seg = -(-nd // self.day_segments)
                chunks = []
                for ci, ((_, r0, c0), (addr, csize, _)) in enumerate(info.items()):
                    if csize != size:
                        raise IOError("%s: unexpected chunk size %d" % (fpath, csize))
                    view = stg[ci * size:(ci + 1) * size]
                    dst = view.view(np.int16).reshape(nd, cy, cx)
                    chunks.append((addr, view, [self.pool.submit(self._gather, dst, src, d0, min(nd, d0 + seg), r0, c0)
                                                for d0 in range(0, nd, seg)]))
                writes.append(self.write_pool.submit(self._write_file, fd, chunks))
            for var in self.variables:
                if var not in present:                          # no daily block came (a normals-only run): the prepared file's chunks
                    os.close(prepared[var][1])                  # hold zeros, not the fill value -- a plain file takes its place
                    os.remove(prepared[var][0])
                    if var in have:
                        self.writer._create(prepared[var][0], tile_id, var, self.days).close()
            for var in have:                                    # the small variables, through the library, meanwhile
                ds = open_dataset(self.writer.fpath(tile_id, var), "a")
                try:
                    self._small(ds, var, arrays)
                finally:
                    ds.close()
            for w in writes:
                w.result()                                      # (raises what a worker raised)
            for stg in stages:
                self._staging.put(stg)
        self._add(copy_s=_t.perf_counter() - t1)
        for var in present:
            nb = (arrays["daily_" + var].nbytes if arrays.get("daily_" + var) is not None
                  else 2 * self.days.size * self.info.tile_size_y * self.info.tile_size_x)
            self._add(int16_bytes=int(nb), disk_bytes=os.path.getsize(self.writer.fpath(tile_id, var)))
        self._add(tiles=1, total_s=_t.perf_counter() - t_begin)
        if k in self.verify:
            for var in present:
                ds = open_dataset(self.writer.fpath(tile_id, var), "r")
                try:
                    v = ds.variables[var]
                    want = arrays.get("daily_" + var)
                    if want is None:                            # deflated on the GPU: what zlib makes of the streams handed over
                        want = self._inflate_tile(arrays["deflated_" + var], v.shape, cy, cx)
                    for r0 in range(0, v.shape[1], cy):        # chunk-aligned slabs: every stored byte is read once
                        if not np.array_equal(v[:, r0:r0 + cy, :], want[:, r0:r0 + cy, :]):
                            raise IOError("%s: read-back differs from what was written" % ds.path)
                    for name, key in ((var + "_normal", "norm_" + var), (var + "_se", "se_" + var), ("inconsist_tair", "ninvalid")):
                        if not np.array_equal(ds.variables[name][:], arrays[key]):
                            raise IOError("%s: %s read-back differs" % (ds.path, name))
                finally:
                    ds.close()
            self._add(verified=1)

    def close(self):
        for futs in self._ahead.values():                       # files prepared for tiles that never came hold no data: remove them
            for f in futs.values():
                try:
                    fpath, fd = f.result()[:2]
                    if fd is not None:
                        os.close(fd)
                    os.remove(fpath)
                    os.rmdir(os.path.dirname(fpath))            # (the tile's directory, when this was its last file)
                except Exception:                               # noqa: BLE001 -- a failed preparation, a directory still in use
                    pass
        self._ahead.clear()
        self.prep_pool.shutdown(wait=True)
        self.write_pool.shutdown(wait=True)
        self.pool.shutdown(wait=True)


# ---- mosaic / monthly product files (tiling.py:567-971,973-1078) --------------------------------------------------------
_PRODUCT_COMMENT = ("The TopoWx ('Topography Weather') gridded dataset contains daily 30-arcsec resolution (~800-m "
                    "resolution; WGS84) interpolations of minimum and maximum topoclimatic air temperature for the "
                    "conterminous U.S. Using both DEM-based variables and MODIS land skin temperature as predictors of air "
                    "temperature, interpolation procedures include moving window regression kriging and geographically "
                    "weighted regression. To avoid artificial climate trends, all input station data are homogenized using "
                    "the GHCN/USHCN Pairwise Homogenization Algorithm "
                    "(http://www.ncdc.noaa.gov/oa/climate/research/ushcn/#phas).")
MOSAIC_UNITS = "days since 1948-1-1 0:0:0"           # tiling.py:699,875,1007


def _product_attrs(ds, title, ds_version_str, comment_prefix=""):
    ds.title = title
    ds.institution = "Pennsylvania State University"
    ds.source = TWX_SOURCE
    ds.history = "Created on: %s , dataset version %s" % (_dt.date.today().strftime("%Y-%m-%d"), ds_version_str)
    ds.references = ("http://dx.doi.org/10.1002/joc.4127 , http://dx.doi.org/10.1002/2014GL062803 , "
                     "http://dx.doi.org/10.1175/JAMC-D-15-0276.1")
    ds.comment = comment_prefix + _PRODUCT_COMMENT
    ds.license = ("Creative Commons Attribution-ShareAlike 4.0 International License "
                  "(http://creativecommons.org/licenses/by-sa/4.0/)")
    ds.Conventions = "CF-1.6"


def _mosaic_chunks(ny, nx):
    """(1, 325, 700) in the reference (tiling.py:720,899,1022: a tenth of the CONUS grid); clipped to the grid."""
    return (1, min(325, ny), min(700, nx))


def create_dly_mosaic_ds(fpath_out, varname, days, lon, lat, ds_version_str, format=None, zlib=True):
    """One year's daily mosaic file as ``create_dly_ann_mosaics`` lays it out (tiling.py:650-736): returns the open
    dataset; the int16 variable is filled with ``_FillValue``."""
    ds = open_dataset(fpath_out, "w", format)
    d0, d1 = _date(days, 0), _date(days, days.size - 1)
    _product_attrs(ds, "Daily Interpolated Topoclimatic Temperature %d-%d" % (_ymd(d0), _ymd(d1)), ds_version_str)
    ds.createDimension("lon", int(np.size(lon)))
    ds.createDimension("lat", int(np.size(lat)))
    ds.createDimension("time", int(days.size))
    ds.createDimension("nv", 2)
    _lonlat_vars(ds, lon, lat)
    _crs_var(ds)
    _time_vars(ds, days, MOSAIC_UNITS)
    long_name, units, std_name, cell_method = VAR_ATTRS[varname]
    v = ds.createVariable(varname, "i2", ("time", "lat", "lon"), chunksizes=_mosaic_chunks(np.size(lat), np.size(lon)),
                          fill_value=FILL_I2, zlib=zlib)
    v.long_name, v.units, v.standard_name = long_name, units, std_name
    v.scale_factor = SCALE_FACTOR
    v.cell_methods = "area: mean time: " + cell_method
    _gridded(v)
    return ds


def create_normals_mosaic_ds(fpath_out, varname, lon, lat, ds_version_str, format=None, zlib=True):
    """The normals mosaic file of ``create_normals_mosaic`` (tiling.py:815-925): packed int16 normals and SE."""
    ds = open_dataset(fpath_out, "w", format)
    _product_attrs(ds, "Interpolated 1981-2010 Monthly Normals for Topoclimatic Temperature", ds_version_str,
                   "1981-2010 monthly normals for the daily TopoWx product. ")
    ds.createDimension("lon", int(np.size(lon)))
    ds.createDimension("lat", int(np.size(lat)))
    ds.createDimension("time", 12)
    ds.createDimension("nv", 2)
    _lonlat_vars(ds, lon, lat)
    _crs_var(ds)
    tm = ds.createVariable("time", "f8", ("time",))
    tm.long_name, tm.units, tm.standard_name, tm.calendar = "time", MOSAIC_UNITS, "time", "standard"
    tm.climatology = "climatology_bounds"
    tm.comment = "Time dimension for the 1981-2010 monthly normals"
    _climatology(ds, "time", _parse_units(MOSAIC_UNITS))
    long_name, units, std_name, cell_method = VAR_ATTRS[varname]
    chunks = _mosaic_chunks(np.size(lat), np.size(lon))
    nv = ds.createVariable(varname + "_normal", "i2", ("time", "lat", "lon"), chunksizes=chunks, fill_value=FILL_I2, zlib=zlib)
    nv.long_name, nv.units, nv.standard_name = "normal " + long_name, units, std_name
    nv.ancillary_variables = varname + "_se"
    nv.comment = "1981-2010 monthly normals"
    nv.scale_factor = SCALE_FACTOR
    nv.cell_methods = "time: %s within years time: mean over years" % cell_method
    _gridded(nv)
    sv = ds.createVariable(varname + "_se", "i2", ("time", "lat", "lon"), chunksizes=chunks, fill_value=FILL_I2, zlib=zlib)
    sv.long_name = long_name + " kriging standard error"
    sv.standard_name, sv.units = "air_temperature standard_error", units
    sv.comment = "Uncertainty in the 1981-2010 monthly normals"
    sv.scale_factor = SCALE_FACTOR
    _gridded(sv)
    return ds


def create_ds_mthly(ds_dly, fpath_out, yr, varname, ds_version_str, format=None, zlib=True):
    """``_create_ds_mthly`` (tiling.py:996-1078): the monthly file of one year's daily mosaic -- lon / lat / crs copied
    from it, a 12-step time axis with month bounds, the int16 variable with the daily variable's attributes."""
    ds = open_dataset(fpath_out, "w", format)
    var_tair = ds_dly.variables[varname]
    lon, lat = ds_dly.variables["lon"][:], ds_dly.variables["lat"][:]
    ds.createDimension("lon", int(lon.size))
    ds.createDimension("lat", int(lat.size))
    ds.createDimension("time", 12)
    ds.createDimension("nv", 2)
    for name, data in (("lon", lon), ("lat", lat), ("crs", None)):
        src = ds_dly.variables[name]
        v = ds.createVariable(name, src.dtype, src.dimensions)
        for a in src.ncattrs():
            if a != "_FillValue":
                v.setncattr(a, src.getncattr(a))
        if data is not None:
            v[:] = data
    times = ds.createVariable("time", "f8", ("time",))
    times.long_name, times.units, times.standard_name, times.calendar = "time", MOSAIC_UNITS, "time", "standard"
    times.bounds = "time_bnds"
    tb = ds.createVariable("time_bnds", "f8", ("time", "nv"))
    d0 = _parse_units(MOSAIC_UNITS)
    edges = [_dt.date(yr, m, 1) for m in range(1, 13)] + [_dt.date(yr + 1, 1, 1)]
    for i in range(12):
        mid = edges[i] + _dt.timedelta(days=(edges[i + 1] - edges[i]).days / 2.0)     # tiling.py:1015-1017
        times[i] = _num(_dt.date(mid.year, mid.month, mid.day), d0)
        tb[i, :] = [_num(edges[i], d0), _num(edges[i + 1], d0)]
    fill = var_tair.getncattr("_FillValue") if "_FillValue" in var_tair.ncattrs() else FILL_I2
    v = ds.createVariable(varname, "i2", ("time", "lat", "lon"), zlib=zlib, chunksizes=_mosaic_chunks(lat.size, lon.size),
                          fill_value=fill)
    for a in var_tair.ncattrs():
        if a not in ("_FillValue", "cell_methods", "_Storage", "_ChunkSizes", "_DeflateLevel", "_Shuffle", "_Endianness"):
            v.setncattr(a, var_tair.getncattr(a))
    v.cell_methods = {"tmin": "time: minimum within days time: mean over days area: mean",
                      "tmax": "time: maximum within days time: mean over days area: mean"}[varname]
    _product_attrs(ds, "Monthly Interpolated Topoclimatic Temperature for %d" % yr, ds_version_str,
                   "Monthly aggregation of the daily TopoWx product.")
    return ds


# ---- serially-complete station database (station_data.py:547-616) -------------------------------------
def _write_ids(ds, dim, name, ids):
    """A station-id coordinate: variable-length strings in NetCDF-4 (create_db_all_stations.py:274), a char array
    ``(dim, string<N>)`` in the classic format."""
    ids = np.asarray(ids)
    if _is_nc4(ds):
        v = ds.createVariable(name, str, (dim,))
        if ids.size:
            v[:] = ids.astype(object)
        return v
    idlen = max([1] + [len(s) for s in ids])
    ds.createDimension("string%d" % idlen, idlen)
    v = ds.createVariable(name, "S1", (dim, "string%d" % idlen))
    if ids.size:
        v[:] = np.array([list(s.ljust(idlen, "\0")) for s in ids], "S1")
    return v


def _read_ids(v):
    raw = v[:]
    if raw.dtype == object:
        return np.array([str(s) for s in raw.ravel()]) if raw.size else np.array([], "U1")
    if raw.ndim == 2:                                                           # chartostring
        return np.array([b"".join(r).rstrip(b"\0 ").decode() for r in raw]) if len(raw) else np.array([], "U1")
    return np.array([s.rstrip(b"\0 ").decode() for s in raw.ravel()])


def _masked_to_nan(v, a):
    """netCDF4-python's auto-mask (``set_auto_maskandscale(True)``, station_data.py:148-164): ``missing_value`` /
    ``_FillValue`` -- or the type's default fill value when the variable has neither -- read back as NaN."""
    a = np.asarray(a, np.float64).copy()
    marks = [float(v.getncattr(att)) for att in ("missing_value", "_FillValue") if att in v.ncattrs()]
    if not marks:
        key = np.dtype(v.dtype).str[1:]
        if key in DEFAULT_FILLS:
            marks = [float(DEFAULT_FILLS[key])]
    for m in marks:
        a[a == m] = np.nan
    return a


def write_station_db(path, stn_da, format=None, zlib=False, obs_chunk_cols=50):
    """A ``StationSerialDataDb`` in the reference's layout (create_db_all_stations.py:262-311, station_data.py:554-616):
    dimensions ``station_id`` and ``time``; one variable per station-table column on ``(station_id,)``; the
    observation variable on ``(time, station_id)`` (NETCDF4: chunked ``(ndays, obs_chunk_cols)`` like the reference's
    ``(ndays, NCDF_CHK_COLS)``)."""
    stns, days = stn_da.stns, stn_da.days
    ds = open_dataset(path, "w", format)
    try:
        n = stns.size
        ds.createDimension(sdb.STN_ID, n)
        ds.createDimension("time", int(days.size))
        _write_ids(ds, sdb.STN_ID, sdb.STN_ID, stns[sdb.STN_ID])
        d0 = _date(days, 0)
        tv = ds.createVariable("time", "f8", ("time",))
        tv.units, tv.calendar, tv.standard_name = _units(d0), "standard", "time"
        tv[:] = np.arange(days.size, dtype=np.float64)
        for name in stns.dtype.names:
            if name == sdb.STN_ID:
                continue
            if stns.dtype[name].kind in "US":
                if _is_nc4(ds):
                    ds.createVariable(name, str, (sdb.STN_ID,))[:] = stns[name].astype(object)
                continue
            v = ds.createVariable(name, "f8", (sdb.STN_ID,), fill_value=FILL_F8)
            v.missing_value = FILL_F8
            v[:] = np.where(np.isnan(stns[name]), FILL_F8, stns[name])
        if stn_da.var is not None:
            kw = {}
            if _is_nc4(ds) and n:
                kw = dict(chunksizes=(int(days.size), min(int(obs_chunk_cols), n)), zlib=zlib)
            ov = ds.createVariable(stn_da.var_name, "f4", ("time", sdb.STN_ID), **kw)
            ov.units = "C"
            ov[:] = stn_da.var
    finally:
        ds.close()


def create_quick_db(path, stns, days, variables, format=None):
    """``create_quick_db`` (create_db_all_stations.py:233-316; step24:87): a station database for a set of stations, a
    period and a list of ``(name, dtype, fill value, long name, units)`` observation variables on ``(time, station_id)``
    -- chunked ``(ndays, 1)`` and compressed in NetCDF-4 -- pre-filled with their fill values."""
    ds = open_dataset(path, "w", format)
    try:
        ds.title = "Weather Station Database"
        ds.institution = "Pennsylvania State University"
        ds.history = "Created on: " + _dt.date.today().strftime("%Y-%m-%d")
        ds.createDimension("time", int(days.size))
        ds.createDimension(sdb.STN_ID, int(stns.size))
        d0 = _date(days, 0)
        tv = ds.createVariable("time", "f8", ("time",))
        tv.long_name, tv.units, tv.standard_name, tv.calendar = "time", _units(d0), "time", "standard"
        tv[:] = np.arange(days.size, dtype=np.float64)
        ids = _write_ids(ds, sdb.STN_ID, sdb.STN_ID, stns[sdb.STN_ID])
        ids.long_name = ids.standard_name = "station id"
        if _is_nc4(ds):
            for name, std in (("station_name", "name"), ("state", "state")):
                if name in stns.dtype.names:
                    v = ds.createVariable(name, str, (sdb.STN_ID,))
                    v.long_name, v.standard_name = name.replace("_", " "), std
                    v[:] = stns[name].astype(object)
        for name, units, std in ((sdb.LAT, "degrees_north", "latitude"), (sdb.LON, "degrees_east", "longitude"),
                                 (sdb.ELEV, "m", "elevation")):
            v = ds.createVariable(name, "f8", (sdb.STN_ID,), fill_value=-9999.0)      # MISSING (create_db_all_stations.py:45)
            v.long_name, v.units, v.standard_name = std, units, std
            v[:] = np.where(np.isnan(stns[name]), -9999.0, stns[name])
        for varname, dtype, fill_value, long_name, units in variables:
            kw = dict(chunksizes=(int(days.size), 1), zlib=True) if _is_nc4(ds) and stns.size else {}
            v = ds.createVariable(varname, dtype, ("time", sdb.STN_ID), fill_value=fill_value, **kw)
            v.long_name, v.units = long_name, units
    finally:
        ds.close()
    return path


def read_station_db(path, var_name, cls=None):
    """``StationSerialDataDb(nc_path, var_name)`` (station_data.py:554-616) on either container (the constructors of
    ``stationdb`` take the path themselves: ``StationDataWrkChk(path, 'tmin')`` as in step25:53-54)."""
    cls = sdb.StationSerialDataDb if cls is None else cls
    return cls(*read_station_db_arrays(path, var_name))


def read_station_db_arrays(path, var_name):
    """(stns, var_name, days, obs) of a station database: what ``_build_stn_struct`` and
    ``StationSerialDataDb.__init__`` read (station_data.py:126-183,554-616) -- every ``(station_id,)`` variable
    (numbers with masked entries as NaN, strings as they are), ``(station_id, string*)`` char arrays, the day axis and
    the ``(time, station_id)`` observations."""
    if not os.path.exists(path):
        raise IOError("no such station database: %s" % path)
    ds = open_dataset(path, "r", rdcc_nbytes=256 << 20)
    try:
        days = days_of(ds)
        ids = _read_ids(ds.variables[sdb.STN_ID])
        cols = []
        for name, v in ds.variables.items():
            if name == sdb.STN_ID:
                continue
            dims = tuple(v.dimensions)
            chara = len(dims) == 2 and dims[0] == sdb.STN_ID and dims[1].startswith("string")
            if dims == (sdb.STN_ID,) and np.dtype(v.dtype).kind in "fiu":
                cols.append((name, np.float64, _masked_to_nan(v, v[:])))
            elif chara or (dims == (sdb.STN_ID,) and np.dtype(v.dtype).kind in "OS"):
                s = _read_ids(v)
                cols.append((name, "U%d" % max(1, max([len(x) for x in s] + [1])), s))
        dt = [(sdb.STN_ID, "U%d" % max(1, max([len(s) for s in ids] + [1])))] + [(k, t) for k, t, _ in cols]
        stns = np.empty(ids.size, dtype=dt)
        stns[sdb.STN_ID] = ids
        for k, _, a in cols:
            stns[k] = a
        obs = None
        if var_name in ds.variables:
            obs = np.ascontiguousarray(ds.variables[var_name][:], np.float32)
    finally:
        ds.close()
    return stns, var_name, days, obs


def convert_station_db(src, dst, var_name, format):
    """Rewrite a station database in the other container (e.g. NetCDF-4 -> classic for a machine without libhdf5)."""
    write_station_db(dst, sdb.StationSerialDataDb(*read_station_db_arrays(src, var_name)), format=format)
    return dst


# ---- per-climate-division cross-validation MAE files (optimize.py:39-82, step21:66-128) ------------------
def climdiv_optim_nstns_path(path_out, tair_var, climdiv):
    """``optim_nstns_<var>_climdiv<id>.nc`` (optimize.py:61, :302, :356)."""
    return os.path.join(path_out, "optim_nstns_%s_climdiv%d.nc" % (tair_var, int(climdiv)))


def create_climdiv_optim_nstns_db(path_out, tair_var, stn_ids, nstns_rng, climdiv, format=None):
    """``create_climdiv_optim_nstns_db`` (optimize.py:39-82) with the reference's signature: creates
    ``optim_nstns_<var>_climdiv<id>.nc`` -- dimensions ``min_nghs``, ``stn_id``, ``mth``; variables ``min_nghs`` i4,
    ``mth`` i4 = 1..12, ``stn_id`` and ``mae`` f8 ``(mth, min_nghs, stn_id)`` with ``missing_value`` = the f8 default
    fill -- and returns the OPEN dataset for the writer rank's ``ds.variables['mae'][:, :, x] = np.abs(err)``
    (step21:124-128)."""
    stn_ids = np.asarray(stn_ids)
    nstns_rng = np.asarray(nstns_rng, np.int32)
    ds = open_dataset(climdiv_optim_nstns_path(path_out, tair_var, climdiv), "w", format)
    ds.title = "Cross Validation MAE for Different N Neighboring Stations: " + tair_var
    ds.institution = "University of Montana Numerical Terradynamics Simulation Group"
    ds.history = "Created on: " + _dt.date.today().strftime("%Y-%m-%d")
    ds.createDimension("min_nghs", int(nstns_rng.size))
    ds.createDimension("stn_id", int(stn_ids.size))
    ids = _write_ids(ds, "stn_id", "stn_id", stn_ids)
    ids.long_name = ids.standard_name = "station id"
    ng = ds.createVariable("min_nghs", "i4", ("min_nghs",))
    ng.long_name = ng.standard_name = "min_nghs"
    ng[:] = nstns_rng
    ds.createDimension("mth", 12)
    mv = ds.createVariable("mth", "i4", ("mth",))
    mv[:] = np.arange(1, 13, dtype=np.int32)
    v = ds.createVariable("mae", "f8", ("mth", "min_nghs", "stn_id"), fill_value=FILL_F8)
    v.long_name, v.units, v.standard_name = "mean absolute error", "C", "mean_absolute_error"
    v.missing_value = FILL_F8
    ds.sync()
    return ds


def write_climdiv_optim_nstns_db(path_out, tair_var, stn_ids, nstns_rng, climdiv, mae, format=None):
    """``create_climdiv_optim_nstns_db`` + the writer rank's assignments in one call: ``mae`` is
    ``[12, n_bandwidths, n_stations]``; a station that was never written (NaN here) holds the fill value, as a
    never-written slot of the reference's file does."""
    stn_ids = np.asarray(stn_ids)
    mae = np.asarray(mae, np.float64)
    if mae.shape != (12, np.size(nstns_rng), stn_ids.size):
        raise ValueError("mae must be [12, n_bandwidths, n_stations]")
    ds = create_climdiv_optim_nstns_db(path_out, tair_var, stn_ids, nstns_rng, climdiv, format)
    try:
        if mae.size:
            ds.variables["mae"][:] = np.where(np.isnan(mae), FILL_F8, mae)
    finally:
        ds.close()
    return climdiv_optim_nstns_path(path_out, tair_var, climdiv)


def read_climdiv_optim_nstns_db(fpath):
    """What ``set_optim_nstns_tair_norm / _anom`` read from a division's file (optimize.py:304-307): ``(mae[12, nb, n]``
    with fill / missing values as NaN -- netCDF4's auto-mask --, ``min_nghs[nb]``, ``stn_ids[n])``."""
    ds = open_dataset(fpath, "r")
    try:
        v = ds.variables["mae"]
        mae = _masked_to_nan(v, v[:])
        nghs = np.asarray(ds.variables["min_nghs"][:], np.int32).copy()
        ids = _read_ids(ds.variables["stn_id"])
    finally:
        ds.close()
    return mae, nghs, ids


# ---- containers: what is read directly, and the classic fallback ----------------------------------------------------
CONVERT_HELP = """\
TopoWx station databases (NetCDF-4 / HDF5) and topowx_amd
=========================================================
topowx_amd.ncio opens a serially-complete TopoWx database written by the reference (create_db_all_stations.py /
infill: NetCDF-4, zlib-chunked, variable-length string ids) DIRECTLY: topowx_amd.h5nc binds the HDF5 library
(libhdf5 + libhdf5_hl, found on the loader path, in /opt/conda/lib or in $TWX_HDF5_LIBDIR) with ctypes and reads what
_build_stn_struct and StationSerialDataDb.__init__ read (station_data.py:126-183,554-616):

    StationDataWrkChk("stns_tmin.nc", "tmin")             # step25:53-54, unchanged call site
    python -m topowx_amd.ncio --check stns_tmin.nc tmin   # lists what was read

What is read: dimension station_id, time (daily, gap-free; units "days since YYYY-MM-DD ..."); every variable shaped
(station_id,): longitude, latitude, elevation, tdi, mask, bad, climdiv, and for MM = 01..12 lstMM, normMM,
optim_nnghsMM, optim_nnghs_anomMM, vario_nugMM, vario_psillMM, vario_rngMM, with _FillValue / missing_value (or the
type's default fill value) read back as NaN; string columns (station_name, state); the observation variable tmin /
tmax shaped (time, station_id), float32.

Only a machine WITHOUT the HDF5 library needs a conversion to classic netCDF (NetCDF-3, 64-bit offset, through
scipy.io.netcdf_file: station ids become a fixed-width char array station_id(station_id, string16)).  Run once, where
libhdf5 is available:

    python -m topowx_amd.ncio --convert in.nc stns_tmin.nc tmin NETCDF3_64BIT

or with the netCDF tools: rewrite the string ids as a char array (netCDF4.stringtochar), then
nccopy -k 64-bit-offset tmp.nc stns_tmin.nc.
"""


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m topowx_amd.ncio", description="netCDF containers of topowx_amd")
    ap.add_argument("--convert-help", action="store_true", help="which containers are read directly, and the classic fallback")
    ap.add_argument("--check", nargs=2, metavar=("PATH", "VAR"), help="open a station database and list what was read")
    ap.add_argument("--convert", nargs=4, metavar=("SRC", "DST", "VAR", "FORMAT"), help="rewrite a station database in FORMAT")
    args = ap.parse_args(argv)
    if args.convert:
        convert_station_db(args.convert[0], args.convert[1], args.convert[2], args.convert[3])
        args.check = [args.convert[1], args.convert[2]]
    if args.convert_help or not args.check:
        print(CONVERT_HELP)
        return 0
    da = sdb.StationSerialDataDb(args.check[0], args.check[1])
    want = [sdb.LON, sdb.LAT, sdb.ELEV, sdb.TDI, sdb.MASK, sdb.BAD, sdb.CLIMDIV] + [
        namer(m) for _, namer in sdb.MONTHLY_FIELDS for m in range(1, 13)]
    missing = [f for f in want if f not in da.stns.dtype.names]
    print("%s: stations %d, days %d (%s .. %s), obs %s, fields %d, missing fields: %s" % (
        file_format(args.check[0]), da.stns.size, da.days.size, da.days["YMD"][0], da.days["YMD"][-1],
        "none" if da.var is None else str(da.var.shape), len(da.stns.dtype.names), missing or "none"))
    return 1 if missing else 0


if __name__ == "__main__":
    raise SystemExit(main())
