"""NetCDF-4 containers on the HDF5 library itself.

The reference opens its serially-complete station database and writes its tiles / mosaics through
netCDF4-python (``twx/db/station_data.py:554-616``, ``twx/interp/tiling.py:304-537``).  netCDF4-python and
h5py are not in this image, but ``libhdf5`` / ``libhdf5_hl`` (1.10) are, and a NetCDF-4 file IS an HDF5 file
whose dimensions are HDF5 dimension scales.  This module binds the C library with ``ctypes`` and implements
the part of the NetCDF-4 data model the path needs, with netCDF4-python's names:

* ``Dataset(path, mode)``: ``dimensions``, ``variables``, ``createDimension``, ``createVariable(name, dtype,
  dims, fill_value=, chunksizes=, zlib=, complevel=, shuffle=)``, global attributes, ``close``;
* ``Variable``: ``dimensions``, ``shape``, ``dtype``, hyperslab ``__getitem__`` / ``__setitem__`` (unit stride),
  attributes, ``chunking()``, ``filters()``; variable-length strings (``str``), ``S1`` char arrays, the
  integer and floating-point types.

File conventions followed (those of netCDF-C's libhdf5 dispatch layer, so that netCDF-C / netCDF4-python /
GDAL open the files as NetCDF-4): creation-order tracking on the root group and on every dataset; one dataset per
dimension with ``CLASS = "DIMENSION_SCALE"``, ``NAME`` (the variable's name for a coordinate variable, the
``"This is a netCDF dimension but not a netCDF variable."`` marker otherwise) and ``_Netcdf4Dimid``;
``DIMENSION_LIST`` / ``REFERENCE_LIST`` through ``H5DSattach_scale``; text attributes as fixed-length
null-terminated strings on a scalar dataspace, numeric attributes on a 1-D dataspace; ``_FillValue`` both as
attribute and as the dataset's HDF5 fill value; ``_NCProperties`` on the root group.  Reading accepts what
netCDF-C writes: either string flavour, big- or little-endian numbers, ``_Netcdf4Coordinates``, the
``_nc4_non_coord_`` name prefix, files without dimension scales (``phony_dim_<n>`` as netCDF-C names them).

Not supported (not on the path): groups, unlimited dimensions, compound / enum / opaque types, strided or
fancy indexing.  ``available()`` tells whether the library could be loaded; ``topowx_amd.ncio`` falls back to
classic netCDF with a clear message when it cannot.
"""
import ctypes as C
import ctypes.util
import os
import threading
from collections import OrderedDict

import numpy as np

__all__ = ["Dataset", "Variable", "available", "is_hdf5", "library_version", "H5Error"]

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
_LOCK = threading.RLock()
_LIB = None          # (libhdf5, libhdf5_hl) or False

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_CRT_ORDER = 0x0001 | 0x0002
H5S_SCALAR, H5S_SELECT_SET = 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_VLEN = 0, 1, 3, 9
H5T_VARIABLE = C.c_size_t(-1).value
H5D_CHUNKED = 2
H5_INDEX_NAME, H5_INDEX_CRT_ORDER, H5_ITER_INC = 0, 1, 0
H5T_SGN_NONE = 0
H5Z_FILTER_DEFLATE, H5Z_FILTER_SHUFFLE = 1, 2
H5D_ALLOC_TIME_EARLY, H5D_FILL_TIME_NEVER = 1, 1
HADDR_UNDEF = (1 << 64) - 1
# netCDF's default fill values (netcdf.h NC_FILL_*; netCDF4.default_fillvals)
DEFAULT_FILLS = {"i1": -127, "u1": 255, "i2": -32767, "u2": 65535, "i4": -2147483647, "u4": 4294967295,
                 "i8": -9223372036854775806, "u8": 18446744073709551614, "f4": 9.969209968386869e36, "f8": 9.969209968386869e36}
NOT_A_VAR = "This is a netCDF dimension but not a netCDF variable."
NON_COORD = "_nc4_non_coord_"
HIDDEN = {"CLASS", "NAME", "REFERENCE_LIST", "DIMENSION_LIST", "_Netcdf4Dimid", "_Netcdf4Coordinates",
          "_nc3_strict", "_NCProperties", "_Netcdf4BitOffset"}


class H5Error(IOError):
    pass


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def _candidates(stem):
    if os.environ.get("TWX_HDF5_DISABLE"):              # (tests: behave as on a machine without the library)
        return []
    env = os.environ.get("TWX_HDF5_LIBDIR")
    out = []
    if env:
        out += [os.path.join(env, "lib%s.so" % stem)]
    found = ctypes.util.find_library(stem)
    if found:
        out.append(found)
    out += ["lib%s.so" % stem]
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib/x86_64-linux-gnu", "/usr/local/lib"):
        out.append(os.path.join(d, "lib%s.so" % stem))
    return out


def _load():
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        lib = hl = None
        for p in _candidates("hdf5"):
            try:
                lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if lib is not None:
            base = os.path.dirname(getattr(lib, "_name", "") or "")
            for p in ([os.path.join(base, "libhdf5_hl.so")] if base else []) + _candidates("hdf5_hl"):
                try:
                    hl = C.CDLL(p, mode=C.RTLD_GLOBAL)
                    break
                except OSError:
                    continue
        if lib is None or hl is None:
            _LIB = False
            return _LIB
        # the bindings below are those of HDF5 >= 1.10 (64-bit hid_t; the library globals are read as 8 bytes): an older
        # library would hand back truncated ids and garbage type handles -- treat it as absent (classic netCDF fallback)
        try:
            lib.H5get_libversion.restype, lib.H5get_libversion.argtypes = herr_t, [C.POINTER(C.c_uint)] * 3
            a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
            if lib.H5get_libversion(C.byref(a), C.byref(b), C.byref(c)) < 0 or (a.value, b.value) < (1, 10):
                _LIB = False
                return _LIB
        except AttributeError:
            _LIB = False
            return _LIB
        _declare(lib, hl)
        if lib.H5open() < 0:
            _LIB = False
            return _LIB
        lib.H5Eset_auto2(hid_t(0), None, None)          # errors become exceptions here, not prints
        _LIB = (lib, hl)
    return _LIB


def available():
    """True when libhdf5 + libhdf5_hl could be loaded."""
    return bool(_load())


def library_version():
    lib, _ = _need()
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return "%d.%d.%d" % (a.value, b.value, c.value)


def is_hdf5(path):
    """HDF5 signature at offset 0 (what netCDF-C tests first; user blocks are not used by NetCDF-4 writers)."""
    with open(path, "rb") as fh:
        return fh.read(8) == b"\x89HDF\r\n\x1a\n"


def _need():
    lib = _load()
    if not lib:
        raise H5Error("libhdf5 / libhdf5_hl not found (searched the loader path, /opt/conda/lib and $TWX_HDF5_LIBDIR): "
                      "NetCDF-4 containers are unavailable, use the classic-netCDF layout of topowx_amd.ncio")
    return lib


def _declare(lib, hl):
    P, cp, vp = C.POINTER, C.c_char_p, C.c_void_p
    sig = {
        "H5open": (herr_t, []), "H5get_libversion": (herr_t, [P(C.c_uint)] * 3),
        "H5Eset_auto2": (herr_t, [hid_t, vp, vp]),
        "H5Fcreate": (hid_t, [cp, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [cp, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, C.c_int]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
        "H5Pset_link_creation_order": (herr_t, [hid_t, C.c_uint]), "H5Pset_attr_creation_order": (herr_t, [hid_t, C.c_uint]),
        "H5Pset_chunk": (herr_t, [hid_t, C.c_int, P(hsize_t)]), "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, P(hsize_t)]),
        "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pset_shuffle": (herr_t, [hid_t]),
        "H5Pset_fill_value": (herr_t, [hid_t, hid_t, vp]), "H5Pget_layout": (C.c_int, [hid_t]),
        "H5Pget_nfilters": (C.c_int, [hid_t]),
        "H5Pget_filter2": (C.c_int, [hid_t, C.c_uint, P(C.c_uint), P(C.c_size_t), P(C.c_uint), C.c_size_t, cp, P(C.c_uint)]),
        "H5Pset_chunk_cache": (herr_t, [hid_t, C.c_size_t, C.c_size_t, C.c_double]),
        "H5Pset_alloc_time": (herr_t, [hid_t, C.c_int]), "H5Pset_fill_time": (herr_t, [hid_t, C.c_int]),
        "H5Pset_alignment": (herr_t, [hid_t, hsize_t, hsize_t]),
        "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, P(hsize_t), P(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, P(hsize_t), P(hsize_t)]),
        "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, P(hsize_t), P(hsize_t), P(hsize_t), P(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, cp, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, cp, hid_t]),
        "H5Dclose": (herr_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dget_create_plist": (hid_t, [hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]), "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]),
        "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, vp]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tset_strpad": (herr_t, [hid_t, C.c_int]), "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
        "H5Acreate2": (hid_t, [hid_t, cp, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, cp, hid_t]),
        "H5Aclose": (herr_t, [hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, vp]), "H5Aread": (herr_t, [hid_t, hid_t, vp]),
        "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aexists": (C.c_int, [hid_t, cp]),
        "H5Adelete": (herr_t, [hid_t, cp]),
        "H5Aiterate2": (herr_t, [hid_t, C.c_int, C.c_int, P(hsize_t), vp, vp]),
        "H5Gget_info": (herr_t, [hid_t, P(_GInfo)]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, cp, C.c_int, C.c_int, hsize_t, cp, C.c_size_t, hid_t]),
        "H5Ldelete": (herr_t, [hid_t, cp, hid_t]),
        "H5Iget_name": (C.c_ssize_t, [hid_t, cp, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    opt = {   # direct chunk access: H5Dwrite_chunk since 1.10.3, the chunk queries since 1.10.5 (Variable.write_chunk_raw / chunk_info)
        "H5Dwrite_chunk": (herr_t, [hid_t, hid_t, C.c_uint32, P(hsize_t), C.c_size_t, vp]),
        "H5Dget_num_chunks": (herr_t, [hid_t, hid_t, P(hsize_t)]),
        "H5Dget_chunk_info": (herr_t, [hid_t, hid_t, hsize_t, P(hsize_t), P(C.c_uint), P(C.c_uint64), P(hsize_t)]),
    }
    for name, (res, args) in opt.items():
        f = getattr(lib, name, None)
        if f is not None:
            f.restype, f.argtypes = res, args
    hsig = {
        "H5DSset_scale": (herr_t, [hid_t, cp]), "H5DSattach_scale": (herr_t, [hid_t, hid_t, C.c_uint]),
        "H5DSis_scale": (C.c_int, [hid_t]), "H5DSget_num_scales": (C.c_int, [hid_t, C.c_uint]),
        "H5DSiterate_scales": (herr_t, [hid_t, C.c_uint, P(C.c_int), vp, vp]),
        "H5DSis_attached": (C.c_int, [hid_t, hid_t, C.c_uint]),
    }
    for name, (res, args) in hsig.items():
        f = getattr(hl, name)
        f.restype, f.argtypes = res, args


def _g(name):
    """A library global such as H5T_NATIVE_DOUBLE_g (the macros of the C headers resolve to these after H5open)."""
    return hid_t.in_dll(_need()[0], name).value


_NATIVE = {"i1": "H5T_NATIVE_SCHAR_g", "u1": "H5T_NATIVE_UCHAR_g", "i2": "H5T_NATIVE_SHORT_g", "u2": "H5T_NATIVE_USHORT_g",
           "i4": "H5T_NATIVE_INT_g", "u4": "H5T_NATIVE_UINT_g", "i8": "H5T_NATIVE_LLONG_g", "u8": "H5T_NATIVE_ULLONG_g",
           "f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g"}


def _native(dtype):
    dt = np.dtype(dtype)
    key = "%s%d" % (dt.kind, dt.itemsize)
    if key not in _NATIVE:
        raise TypeError("no NetCDF-4 type for numpy dtype %r" % (dt,))
    return _g(_NATIVE[key])


def _chk(rc, what):
    if rc < 0:
        raise H5Error("HDF5 call failed: %s" % what)
    return rc


def _dims(n, vals=None):
    a = (hsize_t * max(n, 1))()
    if vals is not None:
        for i, v in enumerate(vals):
            a[i] = int(v)
    return a


class _Attrs(object):
    """Attribute access shared by ``Dataset`` (root group) and ``Variable`` (dataset)."""

    def _loc(self):
        raise NotImplementedError

    def _attr_names(self):
        lib, _ = _need()
        names = []

        @C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)
        def visit(_loc, name, _info, _data):
            names.append(name.decode())
            return 0
        with _LOCK:
            for idx in (H5_INDEX_CRT_ORDER, H5_INDEX_NAME):
                del names[:]
                n = hsize_t(0)
                if lib.H5Aiterate2(self._loc(), idx, H5_ITER_INC, C.byref(n), visit, None) >= 0:
                    break
        return names

    def ncattrs(self):
        return [a for a in self._attr_names() if a not in HIDDEN]

    def getncattr(self, name):
        lib, _ = _need()
        with _LOCK:
            loc = self._loc()
            if lib.H5Aexists(loc, name.encode()) <= 0:
                raise AttributeError("NetCDF: Attribute not found: %s" % name)
            aid = _chk(lib.H5Aopen(loc, name.encode(), 0), "H5Aopen(%s)" % name)
            t = lib.H5Aget_type(aid)
            sp = lib.H5Aget_space(aid)
            try:
                n = max(int(lib.H5Sget_simple_extent_npoints(sp)), 0)
                cls = lib.H5Tget_class(t)
                if cls == H5T_STRING:
                    if lib.H5Tis_variable_str(t) > 0:
                        buf = (C.c_char_p * max(n, 1))()
                        _chk(lib.H5Aread(aid, t, buf), "H5Aread(%s)" % name)
                        vals = [(buf[i] or b"").decode("utf-8", "replace") for i in range(n)]
                        lib.H5Dvlen_reclaim(t, sp, 0, buf)
                    else:
                        size = lib.H5Tget_size(t)
                        buf = C.create_string_buffer(size * max(n, 1))
                        _chk(lib.H5Aread(aid, t, buf), "H5Aread(%s)" % name)
                        raw = buf.raw
                        vals = [raw[i * size:(i + 1) * size].split(b"\0")[0].decode("utf-8", "replace") for i in range(n)]
                    return vals[0] if n == 1 else (vals if n else "")
                if cls in (H5T_INTEGER, H5T_FLOAT):
                    dt = _np_dtype(lib, t)
                    out = np.empty(max(n, 1), dt)
                    _chk(lib.H5Aread(aid, _native(dt), out.ctypes.data_as(C.c_void_p)), "H5Aread(%s)" % name)
                    return out[0] if n == 1 else out[:n]
                return None
            finally:
                lib.H5Sclose(sp)
                lib.H5Tclose(t)
                lib.H5Aclose(aid)

    def setncattr(self, name, value):
        lib, _ = _need()
        self._writable()
        with _LOCK:
            loc = self._loc()
            if lib.H5Aexists(loc, name.encode()) > 0:
                _chk(lib.H5Adelete(loc, name.encode()), "H5Adelete(%s)" % name)
            if isinstance(value, (str, bytes)):
                raw = value.encode() if isinstance(value, str) else value
                t = lib.H5Tcopy(_g("H5T_C_S1_g"))
                lib.H5Tset_size(t, max(len(raw), 1))
                lib.H5Tset_strpad(t, 0)
                sp = lib.H5Screate(H5S_SCALAR)
                aid = _chk(lib.H5Acreate2(loc, name.encode(), t, sp, 0, 0), "H5Acreate2(%s)" % name)
                buf = C.create_string_buffer(raw, max(len(raw), 1))
                rc = lib.H5Awrite(aid, t, buf)
                lib.H5Aclose(aid), lib.H5Sclose(sp), lib.H5Tclose(t)
                _chk(rc, "H5Awrite(%s)" % name)
                return
            arr = np.atleast_1d(np.asarray(value))
            if arr.dtype.kind == "b":
                arr = arr.astype(np.int8)
            if arr.dtype.kind in "US":
                return self.setncattr(name, " ".join(str(x) for x in arr.tolist()))
            arr = np.ascontiguousarray(arr.ravel())
            t = _native(arr.dtype)
            sp = lib.H5Screate_simple(1, _dims(1, [arr.size]), None)
            aid = _chk(lib.H5Acreate2(loc, name.encode(), t, sp, 0, 0), "H5Acreate2(%s)" % name)
            rc = lib.H5Awrite(aid, t, arr.ctypes.data_as(C.c_void_p))
            lib.H5Aclose(aid), lib.H5Sclose(sp)
            _chk(rc, "H5Awrite(%s)" % name)

    def _set_scalar_int(self, name, value):
        """``_Netcdf4Dimid``: H5T_NATIVE_INT on a scalar dataspace, as netCDF-C writes it."""
        lib, _ = _need()
        with _LOCK:
            loc = self._loc()
            if lib.H5Aexists(loc, name.encode()) > 0:
                lib.H5Adelete(loc, name.encode())
            sp = lib.H5Screate(H5S_SCALAR)
            aid = _chk(lib.H5Acreate2(loc, name.encode(), _g("H5T_NATIVE_INT_g"), sp, 0, 0), "H5Acreate2(%s)" % name)
            v = C.c_int(int(value))
            rc = lib.H5Awrite(aid, _g("H5T_NATIVE_INT_g"), C.byref(v))
            lib.H5Aclose(aid), lib.H5Sclose(sp)
            _chk(rc, "H5Awrite(%s)" % name)

    def _has_attr(self, name):
        lib, _ = _need()
        with _LOCK:
            return lib.H5Aexists(self._loc(), name.encode()) > 0

    # netCDF4-python's attribute syntax: v.units = "C"; v.units
    def __getattr__(self, name):
        if name.startswith("_") and name not in ("_FillValue",):
            raise AttributeError(name)
        return self.getncattr(name)

    def __setattr__(self, name, value):
        if name in self._OWN or (name.startswith("_") and name != "_FillValue"):
            object.__setattr__(self, name, value)
        else:
            self.setncattr(name, value)


def _np_dtype(lib, t):
    cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
    if cls == H5T_INTEGER:
        return np.dtype("%s%d" % ("u" if lib.H5Tget_sign(t) == H5T_SGN_NONE else "i", size))
    if cls == H5T_FLOAT:
        return np.dtype("f%d" % size)
    raise TypeError("not a numeric HDF5 type")


class Variable(_Attrs):
    _OWN = {"name", "dimensions", "shape", "dtype", "ndim", "size"}

    def __init__(self, ds, name, did, dimensions, shape, dtype, kind):
        object.__setattr__(self, "_ds", ds)
        object.__setattr__(self, "_did", did)
        object.__setattr__(self, "_kind", kind)          # "num" | "vstr" | "fstr"
        self.name, self.dimensions, self.shape, self.dtype = name, tuple(dimensions), tuple(int(s) for s in shape), dtype
        self.ndim, self.size = len(self.shape), int(np.prod(self.shape, dtype=np.int64)) if shape else 1

    def _loc(self):
        if self._ds._fid is None:
            raise H5Error("dataset is closed")
        return self._did

    def _writable(self):
        self._ds._writable()

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a scalar variable")
        return self.shape[0]

    def chunking(self):
        """'contiguous' or the chunk shape (netCDF4-python's ``Variable.chunking``)."""
        lib, _ = _need()
        with _LOCK:
            p = lib.H5Dget_create_plist(self._loc())
            try:
                if lib.H5Pget_layout(p) != H5D_CHUNKED:
                    return "contiguous"
                d = _dims(max(self.ndim, 1))
                lib.H5Pget_chunk(p, max(self.ndim, 1), d)
                return [int(d[i]) for i in range(self.ndim)]
            finally:
                lib.H5Pclose(p)

    def filters(self):
        lib, _ = _need()
        out = {"zlib": False, "shuffle": False, "complevel": 0}
        with _LOCK:
            p = lib.H5Dget_create_plist(self._loc())
            try:
                for i in range(max(lib.H5Pget_nfilters(p), 0)):
                    flags, nel, cfg = C.c_uint(), C.c_size_t(8), C.c_uint()
                    cd = (C.c_uint * 8)()
                    fid = lib.H5Pget_filter2(p, i, C.byref(flags), C.byref(nel), cd, 0, None, C.byref(cfg))
                    if fid == H5Z_FILTER_DEFLATE:
                        out["zlib"], out["complevel"] = True, int(cd[0])
                    elif fid == H5Z_FILTER_SHUFFLE:
                        out["shuffle"] = True
            finally:
                lib.H5Pclose(p)
        return out

    # ---- direct chunk access (no netCDF4-python counterpart; HDF5 >= 1.10.5) -----------------------------------------------
    def chunk_info(self):
        """{chunk offset (tuple of element indices): (file address, stored bytes, filter mask)} of the chunks that exist in
        the file.  For a variable created with ``alloc_early`` these are ALL chunks, at their final addresses: once the file
        has been flushed or closed their bytes may be written straight into it (the file has no user block)."""
        lib, _ = _need()
        if getattr(lib.H5Dget_chunk_info, "argtypes", None) is None:
            raise H5Error("this libhdf5 (%s) has no H5Dget_chunk_info (needs >= 1.10.5)" % library_version())
        out = OrderedDict()
        with _LOCK:
            fs = lib.H5Dget_space(self._loc())
            try:
                n = hsize_t()
                _chk(lib.H5Dget_num_chunks(self._loc(), fs, C.byref(n)), "H5Dget_num_chunks(%s)" % self.name)
                off = _dims(max(self.ndim, 1))
                for i in range(int(n.value)):
                    mask, addr, size = C.c_uint(), C.c_uint64(), hsize_t()
                    _chk(lib.H5Dget_chunk_info(self._loc(), fs, hsize_t(i), off, C.byref(mask), C.byref(addr), C.byref(size)),
                         "H5Dget_chunk_info(%s)" % self.name)
                    if addr.value != HADDR_UNDEF:
                        out[tuple(int(off[a]) for a in range(self.ndim))] = (int(addr.value), int(size.value), int(mask.value))
            finally:
                lib.H5Sclose(fs)
        return out

    def write_chunk_raw(self, offset, data, filter_mask=0):
        """``H5Dwrite_chunk``: ``data`` (bytes-like) IS the stored form of the chunk at element offset ``offset`` -- for a
        deflated variable the caller has shuffled and compressed it (``ncio.TileSink`` does that on a thread pool; the
        library call is then a byte copy).  ``filter_mask``: bit i set = filter i of the pipeline was NOT applied."""
        lib, _ = _need()
        self._writable()
        if getattr(lib.H5Dwrite_chunk, "argtypes", None) is None:
            raise H5Error("this libhdf5 (%s) has no H5Dwrite_chunk (needs >= 1.10.3)" % library_version())
        mv = memoryview(data).cast("B")
        buf = (C.c_char * len(mv)).from_buffer_copy(mv) if mv.readonly else (C.c_char * len(mv)).from_buffer(mv)
        with _LOCK:
            _chk(lib.H5Dwrite_chunk(self._loc(), 0, C.c_uint32(filter_mask), _dims(self.ndim, offset), C.c_size_t(len(mv)), buf),
                 "H5Dwrite_chunk(%s)" % self.name)

    # ---- hyperslab selection ------------------------------------------------------------------------------------
    def _select(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [j for j, k in enumerate(key) if k is Ellipsis][0]
            key = key[:i] + (slice(None),) * (self.ndim - (len(key) - 1)) + key[i + 1:]
        if len(key) > self.ndim:
            raise IndexError("too many indices for variable %s%s" % (self.name, self.shape))
        key = key + (slice(None),) * (self.ndim - len(key))
        start, count, squeeze = [], [], []
        for ax, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, slice):
                a, b, st = k.indices(n)
                if st != 1:
                    raise IndexError("only unit-stride slices are supported")
                start.append(a), count.append(max(b - a, 0))
            else:
                i = int(k)
                if i < 0:
                    i += n
                if not 0 <= i < n:
                    raise IndexError("index %d out of range for dimension %d of %s" % (int(k), ax, self.name))
                start.append(i), count.append(1), squeeze.append(ax)
        return start, count, tuple(squeeze)

    def _spaces(self, start, count):
        lib, _ = _need()
        if not self.ndim:
            return 0, 0
        fs = lib.H5Dget_space(self._loc())
        _chk(lib.H5Sselect_hyperslab(fs, H5S_SELECT_SET, _dims(self.ndim, start), None, _dims(self.ndim, count), None),
             "H5Sselect_hyperslab")
        ms = lib.H5Screate_simple(self.ndim, _dims(self.ndim, count), None)
        return ms, fs

    def __getitem__(self, key):
        lib, _ = _need()
        start, count, squeeze = self._select(key)
        n = int(np.prod(count, dtype=np.int64)) if count else 1
        with _LOCK:
            if n == 0:
                out = np.empty(count, object if self._kind == "vstr" else self.dtype)
            else:
                ms, fs = self._spaces(start, count)
                try:
                    if self._kind == "vstr":
                        t = lib.H5Dget_type(self._loc())
                        buf = (C.c_char_p * n)()
                        rc = lib.H5Dread(self._loc(), t, ms, fs, 0, buf)
                        if rc >= 0:
                            out = np.array([(buf[i] or b"").decode("utf-8", "replace") for i in range(n)], dtype=object).reshape(count)
                            sp = ms if ms else lib.H5Dget_space(self._loc())
                            lib.H5Dvlen_reclaim(t, sp, 0, buf)
                            if not ms:
                                lib.H5Sclose(sp)
                        lib.H5Tclose(t)
                        _chk(rc, "H5Dread(%s)" % self.name)
                    elif self._kind == "fstr":
                        t = lib.H5Dget_type(self._loc())
                        out = np.empty(count, self.dtype)
                        rc = lib.H5Dread(self._loc(), t, ms, fs, 0, out.ctypes.data_as(C.c_void_p))
                        lib.H5Tclose(t)
                        _chk(rc, "H5Dread(%s)" % self.name)
                    else:
                        out = np.empty(count, self.dtype)
                        _chk(lib.H5Dread(self._loc(), _native(self.dtype), ms, fs, 0, out.ctypes.data_as(C.c_void_p)),
                             "H5Dread(%s)" % self.name)
                finally:
                    if ms:
                        lib.H5Sclose(ms), lib.H5Sclose(fs)
        if squeeze:
            out = out.reshape([c for ax, c in enumerate(count) if ax not in squeeze])
        if out.ndim == 0:
            return out[()]
        return out

    def __setitem__(self, key, value):
        lib, _ = _need()
        self._writable()
        start, count, squeeze = self._select(key)
        n = int(np.prod(count, dtype=np.int64)) if count else 1
        if n == 0:
            return
        kept = [c for ax, c in enumerate(count) if ax not in squeeze]
        with _LOCK:
            ms, fs = self._spaces(start, count)
            try:
                if self._kind == "vstr":
                    vals = np.broadcast_to(np.asarray(value, dtype=object), kept).ravel()
                    raw = [v if isinstance(v, bytes) else str(v).encode() for v in vals]
                    buf = (C.c_char_p * n)(*raw)
                    t = lib.H5Dget_type(self._loc())
                    rc = lib.H5Dwrite(self._loc(), t, ms, fs, 0, buf)
                    lib.H5Tclose(t)
                    _chk(rc, "H5Dwrite(%s)" % self.name)
                elif self._kind == "fstr":
                    arr = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=self.dtype), kept))
                    t = lib.H5Dget_type(self._loc())
                    rc = lib.H5Dwrite(self._loc(), t, ms, fs, 0, arr.ctypes.data_as(C.c_void_p))
                    lib.H5Tclose(t)
                    _chk(rc, "H5Dwrite(%s)" % self.name)
                else:
                    v = np.asarray(value)
                    if np.ma.isMaskedArray(value):
                        fill = self.getncattr("_FillValue") if self._has_attr("_FillValue") else None
                        v = np.ma.filled(value, fill) if fill is not None else np.ma.getdata(value)
                    arr = np.ascontiguousarray(np.broadcast_to(v.astype(self.dtype, copy=False), kept))
                    _chk(lib.H5Dwrite(self._loc(), _native(self.dtype), ms, fs, 0, arr.ctypes.data_as(C.c_void_p)),
                         "H5Dwrite(%s)" % self.name)
            finally:
                if ms:
                    lib.H5Sclose(ms), lib.H5Sclose(fs)


class Dataset(_Attrs):
    """A NetCDF-4 file (root group only).  ``mode``: 'r', 'r+' / 'a' (must exist), 'w' (truncate).  ``alignment``
    (mode 'w'): (threshold, alignment) in bytes -- file objects of at least ``threshold`` bytes start on a multiple of
    ``alignment`` (``H5Pset_alignment``; ``ncio.TileSink`` puts the 126 MB chunks of a tile on page boundaries)."""
    _OWN = {"dimensions", "variables", "path", "mode"}
    data_model = "NETCDF4"

    def __init__(self, path, mode="r", rdcc_nbytes=None, alignment=None):
        lib, _ = _need()
        object.__setattr__(self, "_fid", None)
        self.path, self.mode = os.fspath(path), mode
        self.dimensions, self.variables = OrderedDict(), OrderedDict()
        object.__setattr__(self, "_dimid", OrderedDict())         # dimension name -> netCDF dimid
        object.__setattr__(self, "_scale", {})                    # dimension name -> open dataset id of its scale
        object.__setattr__(self, "_pending", [])                  # variables whose scales are attached at close
        object.__setattr__(self, "_rdcc", rdcc_nbytes)
        with _LOCK:
            if mode == "w":
                fcpl = lib.H5Pcreate(_g("H5P_CLS_FILE_CREATE_ID_g"))
                lib.H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER)
                lib.H5Pset_attr_creation_order(fcpl, H5P_CRT_ORDER)
                fapl = 0
                if alignment is not None:
                    fapl = lib.H5Pcreate(_g("H5P_CLS_FILE_ACCESS_ID_g"))
                    _chk(lib.H5Pset_alignment(fapl, hsize_t(int(alignment[0])), hsize_t(int(alignment[1]))), "H5Pset_alignment")
                fid = lib.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, fcpl, fapl)
                lib.H5Pclose(fcpl)
                if fapl:
                    lib.H5Pclose(fapl)
                _chk(fid, "H5Fcreate(%s)" % self.path)
                object.__setattr__(self, "_fid", fid)
                self.setncattr("_NCProperties", "version=2,hdf5=%s,writer=topowx_amd.h5nc" % library_version())
            elif mode in ("r", "r+", "a"):
                if not os.path.exists(self.path):
                    raise IOError("No such file or directory: %s" % self.path)
                fid = lib.H5Fopen(self.path.encode(), H5F_ACC_RDONLY if mode == "r" else H5F_ACC_RDWR, 0)
                if fid < 0:
                    raise H5Error("%s is not an HDF5 / NetCDF-4 file (or cannot be opened in mode %r)" % (self.path, mode))
                object.__setattr__(self, "_fid", fid)
                self._scan()
            else:
                raise ValueError("mode must be 'r', 'r+', 'a' or 'w'")

    # ---- plumbing ---------------------------------------------------------------------------------------------------
    def _loc(self):
        if self._fid is None:
            raise H5Error("dataset is closed")
        return self._fid

    def _writable(self):
        if self.mode == "r":
            raise H5Error("%s is open read-only" % self.path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _link_names(self):
        lib, _ = _need()
        gi = _GInfo()
        _chk(lib.H5Gget_info(self._fid, C.byref(gi)), "H5Gget_info")
        for idx in (H5_INDEX_CRT_ORDER, H5_INDEX_NAME):
            names = []
            for i in range(int(gi.nlinks)):
                n = lib.H5Lget_name_by_idx(self._fid, b".", idx, H5_ITER_INC, i, None, 0, 0)
                if n < 0:
                    names = None
                    break
                buf = C.create_string_buffer(n + 1)
                lib.H5Lget_name_by_idx(self._fid, b".", idx, H5_ITER_INC, i, buf, n + 1, 0)
                names.append(buf.value.decode())
            if names is not None:
                return names
        raise H5Error("cannot list the root group of %s" % self.path)

    def _open_dataset(self, name):
        lib, _ = _need()
        dapl = 0
        if self._rdcc:
            dapl = lib.H5Pcreate(_g("H5P_CLS_DATASET_ACCESS_ID_g"))
            lib.H5Pset_chunk_cache(dapl, 1009, int(self._rdcc), 0.75)
        did = lib.H5Dopen2(self._fid, name.encode(), dapl)
        if dapl:
            lib.H5Pclose(dapl)
        return did

    def _shape_type(self, did):
        lib, _ = _need()
        sp = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        d = _dims(max(nd, 1))
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sp, d, None)
        lib.H5Sclose(sp)
        shape = tuple(int(d[i]) for i in range(max(nd, 0)))
        t = lib.H5Dget_type(did)
        cls = lib.H5Tget_class(t)
        try:
            if cls == H5T_STRING:
                if lib.H5Tis_variable_str(t) > 0:
                    return shape, np.dtype(object), "vstr"
                return shape, np.dtype("S%d" % lib.H5Tget_size(t)), "fstr"
            if cls in (H5T_INTEGER, H5T_FLOAT):
                return shape, _np_dtype(lib, t), "num"
            return shape, None, None
        finally:
            lib.H5Tclose(t)

    def _scale_names(self, did, axis):
        """Names of the dimension scales attached to ``axis`` of a dataset (DIMENSION_LIST)."""
        lib, hl = _need()
        names = []

        @C.CFUNCTYPE(herr_t, hid_t, C.c_uint, hid_t, C.c_void_p)
        def visit(_did, _dim, scale, _data):
            n = lib.H5Iget_name(scale, None, 0)
            buf = C.create_string_buffer(n + 1)
            lib.H5Iget_name(scale, buf, n + 1)
            names.append(buf.value.decode().rsplit("/", 1)[-1])
            return 0
        if hl.H5DSget_num_scales(did, axis) > 0:
            hl.H5DSiterate_scales(did, axis, None, visit, None)
        return names

    def _scan(self):
        """Rebuild dimensions and variables from the root group's datasets."""
        lib, hl = _need()
        found = []                                                 # (link name, did, shape, dtype, kind, is_scale, NAME)
        for name in self._link_names():
            did = self._open_dataset(name)
            if did < 0:
                continue                                           # a group or a named type: not part of the classic model
            shape, dtype, kind = self._shape_type(did)
            is_scale = hl.H5DSis_scale(did) > 0
            tmp = Variable(self, name, did, (), shape, dtype, kind)
            label = tmp.getncattr("NAME") if is_scale and tmp._has_attr("NAME") else None
            dimid = int(tmp.getncattr("_Netcdf4Dimid")) if tmp._has_attr("_Netcdf4Dimid") else None
            found.append((name, did, shape, dtype, kind, is_scale, label, dimid, tmp))
        # dimensions: every scale, in _Netcdf4Dimid order when the file carries it
        scales = [f for f in found if f[5]]
        scales.sort(key=lambda f: (f[7] is None, f[7] if f[7] is not None else 0))
        for i, f in enumerate(scales):
            self.dimensions[f[0]] = f[2][0] if f[2] else 1
            self._dimid[f[0]] = f[7] if f[7] is not None else i
            self._scale[f[0]] = f[1]
        by_id = {v: k for k, v in self._dimid.items()}
        phony = {}
        for name, did, shape, dtype, kind, is_scale, label, dimid, tmp in found:
            if is_scale and isinstance(label, str) and label.startswith(NOT_A_VAR):
                continue                                           # a dimension without coordinate variable
            if kind is None:
                lib.H5Dclose(did)
                continue                                           # a type outside the supported model
            vname = name[len(NON_COORD):] if name.startswith(NON_COORD) else name
            dims = []
            if is_scale and len(shape) > 1 and tmp._has_attr("_Netcdf4Coordinates"):
                dims = [by_id.get(int(i)) for i in np.atleast_1d(tmp.getncattr("_Netcdf4Coordinates"))]
            elif is_scale:
                dims = [name] + [None] * (len(shape) - 1)
            else:
                dims = [None] * len(shape)
            for ax in range(len(shape)):
                if dims[ax] is None:
                    att = self._scale_names(did, ax)
                    if att:
                        dims[ax] = att[0]
                    else:                                          # no scale: netCDF-C invents one per distinct length
                        key = shape[ax]
                        if key not in phony:
                            phony[key] = "phony_dim_%d" % len(phony)
                            self.dimensions[phony[key]] = key
                        dims[ax] = phony[key]
            self.variables[vname] = Variable(self, vname, did, dims, shape, dtype, kind)

    # ---- definition -----------------------------------------------------------------------------------------------------
    def createDimension(self, name, size):
        self._writable()
        if size is None:
            raise NotImplementedError("unlimited dimensions are not supported")
        if name in self.dimensions:
            raise H5Error("NetCDF: String match to name in use: %s" % name)
        self.dimensions[name] = int(size)
        self._dimid[name] = (max(self._dimid.values()) + 1) if self._dimid else 0
        return name

    def createVariable(self, varname, datatype, dimensions=(), zlib=False, complevel=4, shuffle=True, chunksizes=None,
                       fill_value=None, contiguous=False, alloc_early=False):
        """netCDF4-python's ``createVariable``.  ``fill_value=None``: the type's netCDF default fill is the dataset's HDF5
        fill value (never-written regions read back as it, as through netCDF-C) and no ``_FillValue`` attribute is written;
        ``fill_value=False``: no fill value at all.  ``alloc_early`` (no netCDF4-python counterpart; chunked, unfiltered
        variables): every chunk is allocated at creation and NOT pre-filled -- the caller writes all of them, possibly straight
        into the file at the addresses of ``Variable.chunk_info()`` (``ncio.TileSink``)."""
        lib, hl = _need()
        self._writable()
        if isinstance(dimensions, str):
            dimensions = (dimensions,)
        dimensions = tuple(dimensions)
        if varname in self.variables:
            raise H5Error("NetCDF: String match to name in use: %s" % varname)
        for d in dimensions:
            if d not in self.dimensions:
                raise KeyError("dimension %s not defined" % d)
        shape = tuple(self.dimensions[d] for d in dimensions)
        is_coord = bool(dimensions) and dimensions[0] == varname
        link = varname
        if varname in self.dimensions and not is_coord:
            link = NON_COORD + varname
        with _LOCK:
            if datatype is str or datatype == "str":
                t = lib.H5Tcopy(_g("H5T_C_S1_g"))
                lib.H5Tset_size(t, H5T_VARIABLE)
                dtype, kind, own_t = np.dtype(object), "vstr", True
            elif np.dtype(datatype).kind == "S":
                t = lib.H5Tcopy(_g("H5T_C_S1_g"))
                lib.H5Tset_size(t, np.dtype(datatype).itemsize)
                lib.H5Tset_strpad(t, 0)
                dtype, kind, own_t = np.dtype(datatype), "fstr", True
            else:
                dtype, kind, own_t = np.dtype(datatype), "num", False
                t = _native(dtype)
            dcpl = lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            lib.H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER)
            fill = None
            if fill_value is not None and fill_value is not False and kind == "num":
                fill = np.array([fill_value]).astype(dtype)
                _chk(lib.H5Pset_fill_value(dcpl, t, fill.ctypes.data_as(C.c_void_p)), "H5Pset_fill_value")
            elif fill_value is None and kind == "num" and (dtype.kind + str(dtype.itemsize)) in DEFAULT_FILLS:
                dfl = np.array([DEFAULT_FILLS[dtype.kind + str(dtype.itemsize)]]).astype(dtype)
                _chk(lib.H5Pset_fill_value(dcpl, t, dfl.ctypes.data_as(C.c_void_p)), "H5Pset_fill_value")
            if shape and not contiguous and (chunksizes is not None or zlib):
                if chunksizes is None:
                    chunksizes = _default_chunks(shape, dtype.itemsize if kind != "vstr" else 16)
                if len(chunksizes) != len(shape) or any(int(c) < 1 or int(c) > s for c, s in zip(chunksizes, shape) if s):
                    raise ValueError("NetCDF: Bad chunk sizes %r for shape %r" % (tuple(chunksizes), shape))
                _chk(lib.H5Pset_chunk(dcpl, len(shape), _dims(len(shape), chunksizes)), "H5Pset_chunk")
                if zlib and kind != "vstr":
                    if shuffle and dtype.itemsize > 1:
                        lib.H5Pset_shuffle(dcpl)
                    _chk(lib.H5Pset_deflate(dcpl, int(complevel)), "H5Pset_deflate")
                elif alloc_early and kind == "num":
                    _chk(lib.H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY), "H5Pset_alloc_time")
                    _chk(lib.H5Pset_fill_time(dcpl, H5D_FILL_TIME_NEVER), "H5Pset_fill_time")
            sp = lib.H5Screate_simple(len(shape), _dims(len(shape), shape), None) if shape else lib.H5Screate(H5S_SCALAR)
            if is_coord and varname in self._scale:
                # a dimension-only placeholder of an earlier session is replaced by its coordinate variable
                lib.H5Dclose(self._scale.pop(varname))
                lib.H5Ldelete(self._fid, varname.encode(), 0)
            did = lib.H5Dcreate2(self._fid, link.encode(), t, sp, 0, dcpl, 0)
            lib.H5Sclose(sp), lib.H5Pclose(dcpl)
            if own_t:
                lib.H5Tclose(t)
            _chk(did, "H5Dcreate2(%s)" % link)
            var = Variable(self, varname, did, dimensions, shape, dtype, kind)
            self.variables[varname] = var
            if fill is not None:
                var.setncattr("_FillValue", fill)
            if is_coord:
                _chk(hl.H5DSset_scale(did, varname.encode()), "H5DSset_scale(%s)" % varname)
                var._set_scalar_int("_Netcdf4Dimid", self._dimid[varname])
                if len(dimensions) > 1:
                    var.setncattr("_Netcdf4Coordinates", np.array([self._dimid[d] for d in dimensions], np.int32))
                self._scale[varname] = did
            self._pending.append(var)
        return var

    def _materialise(self):
        """Dimension-only scales + DIMENSION_LIST / REFERENCE_LIST of the variables defined in this session."""
        lib, hl = _need()
        if self.mode == "r" or self._fid is None:
            return
        for name, n in self.dimensions.items():
            if name in self._scale or name.startswith("phony_dim_"):
                continue
            sp = lib.H5Screate_simple(1, _dims(1, [n]), None)
            dcpl = lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            lib.H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER)
            did = lib.H5Dcreate2(self._fid, name.encode(), _g("H5T_IEEE_F32BE_g"), sp, 0, dcpl, 0)
            lib.H5Sclose(sp), lib.H5Pclose(dcpl)
            _chk(did, "H5Dcreate2(dimension %s)" % name)
            _chk(hl.H5DSset_scale(did, ("%s%10d" % (NOT_A_VAR, n)).encode()), "H5DSset_scale(%s)" % name)
            Variable(self, name, did, (name,), (n,), np.dtype("f4"), "num")._set_scalar_int("_Netcdf4Dimid", self._dimid[name])
            self._scale[name] = did
        for var in self._pending:
            if var.dimensions and var.dimensions[0] == var.name:
                continue        # a coordinate variable IS a scale: HDF5 attaches nothing to scales (_Netcdf4Coordinates names its dimensions)
            for ax, d in enumerate(var.dimensions):
                sid = self._scale.get(d)
                if sid is not None and hl.H5DSis_attached(var._did, sid, ax) <= 0:
                    _chk(hl.H5DSattach_scale(var._did, sid, ax), "H5DSattach_scale(%s, %s)" % (var.name, d))
        del self._pending[:]

    def sync(self):
        lib, _ = _need()
        with _LOCK:
            self._materialise()
            if self._fid is not None and self.mode != "r":
                lib.H5Fflush(self._fid, 1)

    def close(self):
        if getattr(self, "_fid", None) is None:
            return
        lib, _ = _need()
        with _LOCK:
            try:
                self._materialise()
            finally:
                ids = {v._did for v in self.variables.values()} | set(self._scale.values())
                for did in ids:
                    lib.H5Dclose(did)
                fid = self._fid
                object.__setattr__(self, "_fid", None)
                _chk(lib.H5Fclose(fid), "H5Fclose(%s)" % self.path)


def _default_chunks(shape, itemsize, target=4 << 20):
    """Chunk shape for a compressed variable without explicit chunk sizes: the whole variable below 4 MiB,
    otherwise leading dimensions are halved until a chunk fits (trailing = fastest-varying dimensions stay whole)."""
    c = [max(int(s), 1) for s in shape]
    ax = 0
    while int(np.prod(c, dtype=np.int64)) * itemsize > target and ax < len(c):
        if c[ax] > 1:
            c[ax] = (c[ax] + 1) // 2
        else:
            ax += 1
    return c
