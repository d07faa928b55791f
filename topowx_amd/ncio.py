"""netCDF containers either side of the path.

The reference reads its serially-complete station DB and writes its tiles through netCDF4-python
(``twx/db/station_data.py:547-666``, ``twx/interp/tiling.py:304-537``).  netCDF4 / HDF5 are not available
here; this module keeps the same dimensions, variables and CF attributes on **NetCDF-3 (64-bit offset)**
through ``scipy.io.netcdf_file`` -- files netCDF4-python, GDAL and ncdump open as they are.  What the
classic format cannot carry is dropped: zlib, chunking, variable-length strings (station ids are
``char[station_id][string]`` arrays, the layout ``_build_stn_struct`` already accepts,
station_data.py:134-157).  ``scipy.io.netcdf_file`` holds a file's variables in memory until it is
closed, so a tile costs its size in RAM while it is written.
"""
import datetime as _dt
import os

import numpy as np
from scipy.io import netcdf_file

from . import stationdb as sdb
from .dates import DAY, MONTH, YEAR, get_days_metadata

__all__ = ["TileWriter", "read_tile", "write_station_db", "read_station_db", "read_station_db_arrays", "read_tile_stores", "CONVERT_HELP",
           "climdiv_optim_nstns_path", "write_climdiv_optim_nstns_db", "read_climdiv_optim_nstns_db"]

FILL_I2 = np.int16(-32767)
FILL_F4 = np.float32(9.969209968386869e36)
FILL_I4 = np.int32(-2147483647)
SCALE_FACTOR = np.float32(0.01)                      # tiling.py:36
# long name, units, standard name, cell method (tiling.py:39-42)
VAR_ATTRS = {"tmin": ("minimum air temperature", "C", "air_temperature", "minimum"),
             "tmax": ("maximum air temperature", "C", "air_temperature", "maximum")}


def _date(days, i):
    return _dt.date(int(days[YEAR][i]), int(days[MONTH][i]), int(days[DAY][i]))


def _num(d, d0):
    return float((d - d0).days)


def _mid(a, b):
    return a + _dt.timedelta(days=((b - a).days // 2))


class TileWriter(object):
    """``TileWriter`` (tiling.py:304-537): one file ``<path_out>/<tile_id>/<tile_id>_<varname>.nc`` per tile
    and variable, created on the first chunk, reopened for every further chunk."""

    def __init__(self, tile_grid_info, path_out):
        t = tile_grid_info
        self.tile_ids, self.tile_rc, self.ntiles = t.tile_ids, t.tile_rc, t.ntiles
        self.lons, self.lats = np.asarray(t.lons, np.float64), np.asarray(t.lats, np.float64)
        self.path_out = path_out
        self.tile_size_y, self.tile_size_x = t.tile_size_y, t.tile_size_x
        self.chk_size_y, self.chk_size_x = t.chk_size_y, t.chk_size_x

    def fpath(self, tile_id, varname):
        return os.path.join(self.path_out, tile_id, "%s_%s.nc" % (tile_id, varname))

    def _create(self, fpath, tile_id, varname, days):
        os.makedirs(os.path.dirname(fpath), exist_ok=True)
        ds = netcdf_file(fpath, "w", version=2, mmap=False)
        d0, d1 = _date(days, 0), _date(days, days.size - 1)
        ds.title = "Daily Interpolated Meteorological Data %d-%d" % (
            d0.year * 10000 + d0.month * 100 + d0.day, d1.year * 10000 + d1.month * 100 + d1.day)
        ds.institution = "University of Montana"
        ds.source = "topowx_amd (MI355X implementation of the TopoWx interpolation path)"
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

        units = "days since %d-%d-%d 0:0:0" % (d0.year, d0.month, d0.day)
        times = ds.createVariable("time", "d", ("time",))
        times.long_name, times.units, times.standard_name = "time", units, "time"
        times.calendar, times.bounds = "standard", "time_bnds"
        nums = np.array([_num(_date(days, i), d0) for i in range(days.size)]) + 0.5
        times[:] = nums
        tb = ds.createVariable("time_bnds", "d", ("time", "nv"))
        tb[:, 0], tb[:, 1] = nums - 0.5, nums + 0.5

        tn = ds.createVariable("time_normals", "d", ("time_normals",))
        tn.long_name, tn.units, tn.standard_name, tn.calendar = "time", units, "time", "standard"
        tn.climatology = "climatology_bounds"
        tn.comment = "Time dimension for the 1981-2010 monthly normals"
        cb = ds.createVariable("climatology_bounds", "d", ("time_normals", "nv"))
        for m in range(1, 13):                                            # tiling.py:412-420
            mn, y1 = (m + 1, 1981) if m != 12 else (1, 1982)
            tn[m - 1] = _num(_mid(_dt.date(1981, m, 1), _dt.date(y1, mn, 1)), d0)
            cb[m - 1, 0] = _num(_dt.date(1981, m, 1), d0)
            cb[m - 1, 1] = _num(_dt.date(2010 if m != 12 else 2011, mn, 1), d0)

        la = ds.createVariable("lat", "d", ("lat",))
        la.long_name, la.units, la.standard_name = "latitude", "degrees_north", "latitude"
        la[:] = lats
        lo = ds.createVariable("lon", "d", ("lon",))
        lo.long_name, lo.units, lo.standard_name = "longitude", "degrees_east", "longitude"
        lo[:] = lons

        crs = ds.createVariable("crs", "h", ())                           # tiling.py:539-546
        crs.grid_mapping_name = "latitude_longitude"
        crs.longitude_of_prime_meridian = 0.0
        crs.semi_major_axis = 6378137.0
        crs.inverse_flattening = 298.257223563

        long_name, vunits, std_name, cell_method = VAR_ATTRS[varname]

        def gridded(v):
            v.coordinates, v.grid_mapping = "lat lon", "crs"

        mv = ds.createVariable(varname, "h", ("time", "lat", "lon"))
        mv._FillValue = FILL_I2
        mv.long_name, mv.units, mv.standard_name = long_name, vunits, std_name
        mv.scale_factor = SCALE_FACTOR
        mv.cell_methods = "area: mean time: " + cell_method
        gridded(mv)
        mv[:] = FILL_I2
        nv = ds.createVariable(varname + "_normal", "f", ("time_normals", "lat", "lon"))
        nv._FillValue = FILL_F4
        nv.long_name, nv.units, nv.standard_name = "normal " + long_name, vunits, std_name
        nv.ancillary_variables = varname + "_se"
        nv.comment = "The 1981-2010 monthly normals"
        nv.cell_methods = "time: %s within years time: mean over years" % cell_method
        gridded(nv)
        nv[:] = FILL_F4
        sv = ds.createVariable(varname + "_se", "f", ("time_normals", "lat", "lon"))
        sv._FillValue = FILL_F4
        sv.long_name = long_name + " kriging standard error"
        sv.standard_name, sv.units = "air_temperature standard_error", vunits
        sv.comment = "The uncertainty in the 1981-2010 monthly normals"
        gridded(sv)
        sv[:] = FILL_F4
        iv = ds.createVariable("inconsist_tair", "i", ("lat", "lon"))
        iv._FillValue = FILL_I4
        iv.long_name, iv.units = "number of days interpolated tmin >= tmax", "days"
        iv.comment = ("The number of days daily tmin/tmax had to be adjusted due to interpolated tmin "
                      "being >= interpolated tmax")
        gridded(iv)
        iv[:] = FILL_I4
        return ds

    def write_tile_chunk(self, tile_id, varname, days, str_row, str_col, daily_vals, mthly_normals,
                         mthly_normals_se, ninvalid):
        """tiling.py:488-537; ``daily_vals`` is the packed int16 block (or None for a normals-only run)."""
        fpath = self.fpath(tile_id, varname)
        ds = netcdf_file(fpath, "a", mmap=False) if os.path.exists(fpath) else self._create(fpath, tile_id, varname, days)
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


def _native(a):
    """netCDF classic data is big-endian: copy into the native byte order."""
    a = np.asarray(a)
    return a.astype(a.dtype.newbyteorder("="))


def read_tile(fpath, varname):
    """Arrays of one tile file: daily int16 (raw), normals / SE f4, ninvalid, lon, lat, time."""
    ds = netcdf_file(fpath, "r", mmap=False)
    try:
        g = ds.variables
        out = {"daily": _native(g[varname][:]), "norm": _native(g[varname + "_normal"][:]),
               "se": _native(g[varname + "_se"][:]), "ninvalid": _native(g["inconsist_tair"][:]),
               "lon": _native(g["lon"][:]), "lat": _native(g["lat"][:]),
               "time": _native(g["time"][:]), "time_units": g["time"].units.decode(),
               "scale_factor": g[varname].scale_factor}
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


# ---- serially-complete station database (station_data.py:547-616) -------------------------------------
def write_station_db(path, stn_da):
    """A ``StationSerialDataDb`` as a classic netCDF file in the reference's layout: dimension
    ``station_id`` (+ ``string<N>`` for the ids), ``time``; one variable per station-table column on
    (station_id,); the observation variable on (time, station_id)."""
    stns, days = stn_da.stns, stn_da.days
    ds = netcdf_file(path, "w", version=2, mmap=False)
    try:
        n = stns.size
        idlen = max(len(s) for s in stns[sdb.STN_ID])
        ds.createDimension(sdb.STN_ID, n)
        ds.createDimension("string%d" % idlen, idlen)
        ds.createDimension("time", int(days.size))
        ids = ds.createVariable(sdb.STN_ID, "c", (sdb.STN_ID, "string%d" % idlen))
        ids[:] = np.array([list(s.ljust(idlen, "\0")) for s in stns[sdb.STN_ID]], "S1")
        d0 = _date(days, 0)
        tv = ds.createVariable("time", "d", ("time",))
        tv.units = "days since %d-%d-%d 0:0:0" % (d0.year, d0.month, d0.day)
        tv.calendar, tv.standard_name = "standard", "time"
        tv[:] = [_num(_date(days, i), d0) for i in range(days.size)]
        for name in stns.dtype.names:
            if name == sdb.STN_ID:
                continue
            v = ds.createVariable(name, "d", (sdb.STN_ID,))
            v.missing_value = float(FILL_F4)
            v[:] = np.where(np.isnan(stns[name]), float(FILL_F4), stns[name])
        if stn_da.var is not None:
            ov = ds.createVariable(stn_da.var_name, "f", ("time", sdb.STN_ID))
            ov.units = "C"
            ov[:] = stn_da.var
    finally:
        ds.close()


def read_station_db(path, var_name, cls=None):
    """``StationSerialDataDb(nc_path, var_name)`` (station_data.py:554-616) on a classic netCDF file (the constructors
    of ``stationdb`` take the path themselves: ``StationDataWrkChk(path, 'tmin')`` as in step25:53-54)."""
    cls = sdb.StationSerialDataDb if cls is None else cls
    return cls(*read_station_db_arrays(path, var_name))


def read_station_db_arrays(path, var_name):
    """(stns, var_name, days, obs) of a classic-netCDF station database: what ``_build_stn_struct`` and
    ``StationSerialDataDb.__init__`` read (station_data.py:126-183,554-616)."""
    if not os.path.exists(path):
        raise IOError("no such station database: %s" % path)
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:3] != b"CDF":
        raise IOError("%s is not a classic / 64-bit-offset netCDF file (magic %r): a NetCDF-4 / HDF5 database must be "
                      "converted once -- python -m topowx_amd.ncio --convert-help" % (path, magic))
    ds = netcdf_file(path, "r", mmap=False)
    try:
        tv = ds.variables["time"]
        units = tv.units.decode()
        if not units.startswith("days since "):
            raise ValueError("time units must be 'days since ...'")
        y, m, d = (int(x) for x in units.split()[2].split("-"))
        t = np.floor(np.asarray(tv[:], np.float64)).astype(np.int64)
        d0 = _dt.date(y, m, d)
        if not np.array_equal(t - t[0], np.arange(t.size)):
            raise ValueError("time axis must be daily and gap-free")
        days = get_days_metadata(d0 + _dt.timedelta(days=int(t[0])), d0 + _dt.timedelta(days=int(t[-1])))
        raw = ds.variables[sdb.STN_ID][:]
        ids = np.array([b"".join(r).rstrip(b"\0 ").decode() for r in raw])      # chartostring
        cols = {}
        for name, v in ds.variables.items():
            if v.dimensions == (sdb.STN_ID,):
                a = np.asarray(v[:], np.float64).copy()
                for att in ("missing_value", "_FillValue"):
                    if hasattr(v, att):
                        a[a == float(getattr(v, att))] = np.nan                # auto-mask -> NaN (:159-164)
                cols[name] = a
        dt = [(sdb.STN_ID, "U%d" % max(1, max(len(s) for s in ids)))] + [(k, np.float64) for k in cols]
        stns = np.empty(ids.size, dtype=dt)
        stns[sdb.STN_ID] = ids
        for k, a in cols.items():
            stns[k] = a
        obs = np.asarray(ds.variables[var_name][:], np.float32).copy() if var_name in ds.variables else None
    finally:
        ds.close()
    return stns, var_name, days, obs


# ---- per-climate-division cross-validation MAE files (optimize.py:39-82, step21:66-128) ------------------
FILL_F8 = 9.969209968386869e36          # netCDF4.default_fillvals['f8'] (create_db_all_stations.py:164)


def climdiv_optim_nstns_path(path_out, tair_var, climdiv):
    """``optim_nstns_<var>_climdiv<id>.nc`` (optimize.py:61, :302, :356)."""
    return os.path.join(path_out, "optim_nstns_%s_climdiv%d.nc" % (tair_var, int(climdiv)))


def write_climdiv_optim_nstns_db(path_out, tair_var, stn_ids, nstns_rng, climdiv, mae):
    """``create_climdiv_optim_nstns_db`` (optimize.py:39-82) + the writer rank's ``ds.variables['mae'][:, :, dim2] =
    np.abs(err)`` (step21:124-128), in one call: dimensions ``min_nghs``, ``stn_id`` (+ ``string<N>``: classic netCDF
    has no variable-length strings), ``mth``; variables ``min_nghs`` i4, ``mth`` i4 = 1..12, ``stn_id`` and
    ``mae`` f8 ``(mth, min_nghs, stn_id)`` with ``missing_value`` = the f8 default fill; a station that was never
    written (NaN here) holds the fill value, as a never-written slot of the reference's file does."""
    stn_ids = np.asarray(stn_ids)
    nstns_rng = np.asarray(nstns_rng, np.int32)
    mae = np.asarray(mae, np.float64)
    if mae.shape != (12, nstns_rng.size, stn_ids.size):
        raise ValueError("mae must be [12, n_bandwidths, n_stations]")
    fpath = climdiv_optim_nstns_path(path_out, tair_var, climdiv)
    ds = netcdf_file(fpath, "w", version=2, mmap=False)
    try:
        ds.title = "Cross Validation MAE for Different N Neighboring Stations: " + tair_var
        ds.institution = "University of Montana Numerical Terradynamics Simulation Group"
        ds.history = "Created on: " + _dt.date.today().strftime("%Y-%m-%d")
        idlen = max([1] + [len(s) for s in stn_ids])
        ds.createDimension("min_nghs", int(nstns_rng.size))
        ds.createDimension("stn_id", int(stn_ids.size))
        ds.createDimension("string%d" % idlen, idlen)
        ds.createDimension("mth", 12)
        ids = ds.createVariable("stn_id", "c", ("stn_id", "string%d" % idlen))
        ids.long_name = ids.standard_name = "station id"
        if stn_ids.size:
            ids[:] = np.array([list(s.ljust(idlen, "\0")) for s in stn_ids], "S1")
        ng = ds.createVariable("min_nghs", "i", ("min_nghs",))
        ng.long_name = ng.standard_name = "min_nghs"
        ng[:] = nstns_rng
        mv = ds.createVariable("mth", "i", ("mth",))
        mv[:] = np.arange(1, 13, dtype=np.int32)
        v = ds.createVariable("mae", "d", ("mth", "min_nghs", "stn_id"))
        v.long_name, v.units, v.standard_name = "mean absolute error", "C", "mean_absolute_error"
        v.missing_value = FILL_F8
        v[:] = np.where(np.isnan(mae), FILL_F8, mae)
    finally:
        ds.close()
    return fpath


def read_climdiv_optim_nstns_db(fpath):
    """What ``set_optim_nstns_tair_norm / _anom`` read from a division's file (optimize.py:304-307): ``(mae[12, nb, n]``
    with fill / missing values as NaN -- netCDF4's auto-mask --, ``min_nghs[nb]``, ``stn_ids[n])``."""
    ds = netcdf_file(fpath, "r", mmap=False)
    try:
        v = ds.variables["mae"]
        mae = np.asarray(v[:], np.float64).copy()
        for att in ("missing_value", "_FillValue"):
            if hasattr(v, att):
                mae[mae == float(getattr(v, att))] = np.nan
        nghs = np.asarray(ds.variables["min_nghs"][:], np.int32).copy()
        raw = ds.variables["stn_id"][:]
        ids = np.array([b"".join(r).rstrip(b"\0 ").decode() for r in raw]) if len(raw) else np.array([], "U1")
    finally:
        ds.close()
    return mae, nghs, ids


# ---- NetCDF-4 / HDF5 databases -------------------------------------------------------------------------------------------
CONVERT_HELP = """\
Converting a TopoWx station database (NetCDF-4 / HDF5) for topowx_amd
====================================================================
topowx_amd.ncio reads and writes the reference's layouts (station_data.py:126-183,547-616; tiling.py:304-537) as
classic netCDF (NetCDF-3, 64-bit offset) through scipy.io.netcdf_file: this image has neither netCDF4-python nor
h5py, and an HDF5 layer is out of scope (DESIGN.md section 7).  A serially-complete TopoWx database written by the
reference (create_db_all_stations.py / infill: NetCDF-4, zlib-chunked, variable-length string ids) is converted ONCE,
on any machine that has the netCDF tools:

1. station ids: NetCDF-3 has no string type.  Rewrite the variable-length string variable `station_id(station_id)`
   as a fixed-width char array `station_id(station_id, string16)` -- with NCO:

       ncap2 -O -s 'sid_chr[$station_id,$string16]=" "; ' in.nc tmp.nc      # or, simpler, with Python + netCDF4:

       import netCDF4, numpy as np
       src = netCDF4.Dataset("in.nc"); dst = netCDF4.Dataset("tmp.nc", "w", format="NETCDF4_CLASSIC")
       ids = np.array(src.variables["station_id"][:], "S16")
       for name, d in src.dimensions.items(): dst.createDimension(name, None if d.isunlimited() else len(d))
       dst.createDimension("string16", 16)
       dst.createVariable("station_id", "S1", ("station_id", "string16"))[:] = netCDF4.stringtochar(ids)
       for name, v in src.variables.items():
           if name == "station_id" or v.dtype == str: continue          # other string columns (names, states) are not read
           o = dst.createVariable(name, v.dtype, v.dimensions, fill_value=getattr(v, "_FillValue", None))
           o.setncatts({k: v.getncattr(k) for k in v.ncattrs() if k != "_FillValue"}); o[:] = v[:]
       dst.close()

2. container:  nccopy -k 64-bit-offset tmp.nc stns_tmin.nc
   (removes chunking / zlib; `nccopy -k classic` also works below 2 GiB per variable).

What must survive the conversion (everything _build_stn_struct and StationSerialDataDb.__init__ read):
  * dimension `station_id`, `time` (daily, gap-free; units "days since YYYY-MM-DD ...")
  * every numeric variable shaped (station_id,): longitude, latitude, elevation, tdi, mask, bad, climdiv, and for
    MM = 01..12: lstMM, normMM, optim_nnghsMM, optim_nnghs_anomMM, vario_nugMM, vario_psillMM, vario_rngMM -- with
    their _FillValue / missing_value attributes (masked entries are read back as NaN, station_data.py:159-164)
  * the observation variable tmin / tmax shaped (time, station_id), float32

Then:  StationDataWrkChk("stns_tmin.nc", "tmin")   # step25:53-54, unchanged call site
Check: python -m topowx_amd.ncio --check stns_tmin.nc tmin
"""


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m topowx_amd.ncio", description="classic-netCDF containers of topowx_amd")
    ap.add_argument("--convert-help", action="store_true", help="how to convert a NetCDF-4 / HDF5 TopoWx database once")
    ap.add_argument("--check", nargs=2, metavar=("PATH", "VAR"), help="open a station database and list what was read")
    args = ap.parse_args(argv)
    if args.convert_help or not args.check:
        print(CONVERT_HELP)
        return 0
    da = sdb.StationSerialDataDb(args.check[0], args.check[1])
    want = [sdb.LON, sdb.LAT, sdb.ELEV, sdb.TDI, sdb.MASK, sdb.BAD, sdb.CLIMDIV] + [
        namer(m) for _, namer in sdb.MONTHLY_FIELDS for m in range(1, 13)]
    missing = [f for f in want if f not in da.stns.dtype.names]
    print("stations %d, days %d (%s .. %s), obs %s, fields %d, missing fields: %s" % (
        da.stns.size, da.days.size, da.days["YMD"][0], da.days["YMD"][-1],
        "none" if da.var is None else str(da.var.shape), len(da.stns.dtype.names), missing or "none"))
    return 1 if missing else 0


if __name__ == "__main__":
    raise SystemExit(main())
