"""Runs under an interpreter that HAS h5py (this image: /opt/conda/bin/python3.9, h5py 3.3 on its own libhdf5): the foreign
HDF5 binding that tests/test_h5py_interop.py uses to look at the files topowx_amd.ncio writes, and to write a station
database the way netCDF-C's libhdf5 layer lays one out.  No import of this repository.

    python h5py_peer.py describe FILE            -> JSON: every dataset's shape / dtype / chunks / filters / fill value /
                                                    dimension scales / attributes / sha256 of its values
    python h5py_peer.py write-station-db FILE    -> a NetCDF-4 station database (create_db_all_stations.py:262-311 layout)
                                                    built with h5py's dimension-scale API; prints what it wrote as JSON
"""
import hashlib
import json
import sys

import h5py
import numpy as np

NOT_A_VAR = "This is a netCDF dimension but not a netCDF variable."


def _jsonable(v):
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    if isinstance(v, np.ndarray):
        if v.dtype.kind in "OSU":
            return [_jsonable(x) for x in v.ravel().tolist()]
        if v.dtype.kind == "V" or v.dtype.names:
            return "<compound>"
        return [float(x) if v.dtype.kind == "f" else int(x) for x in v.ravel()]
    if isinstance(v, np.generic):
        return float(v) if v.dtype.kind == "f" else int(v)
    return v


def describe(path):
    out = {}
    with h5py.File(path, "r") as f:
        out["/attrs"] = {k: _jsonable(v) for k, v in f.attrs.items()}
        for name, d in f.items():
            if not isinstance(d, h5py.Dataset):
                continue
            rec = {"shape": list(d.shape), "dtype": str(d.dtype) if d.dtype.kind != "O" else "vlen-str", "chunks": list(d.chunks) if d.chunks else None,
                   "compression": d.compression, "compression_opts": d.compression_opts, "shuffle": bool(d.shuffle),
                   "is_scale": bool(h5py.h5ds.is_scale(d.id)), "attrs": {}}
            rec["fillvalue"] = _jsonable(np.asarray(d.fillvalue)) if d.dtype.kind in "fiu" else None
            for k, v in d.attrs.items():
                if k in ("DIMENSION_LIST", "REFERENCE_LIST"):
                    rec["attrs"][k] = "<references>"
                else:
                    rec["attrs"][k] = _jsonable(v)
            if not rec["is_scale"] or len(d.shape) > 1:
                rec["dims"] = [[f[r].name.lstrip("/") for r in refs] for refs in (d.attrs["DIMENSION_LIST"] if "DIMENSION_LIST" in d.attrs else [])]
                rec["dims"] = [x[0] if x else None for x in rec["dims"]]
            if d.dtype.kind == "O":
                vals = [x.decode() if isinstance(x, bytes) else str(x) for x in d[...].ravel()]
                rec["sha256"] = hashlib.sha256("\0".join(vals).encode()).hexdigest()
            elif d.shape == ():
                rec["sha256"] = None
            else:
                h = hashlib.sha256()
                step = max(1, (64 << 20) // max(1, int(np.prod(d.shape[1:], dtype=np.int64)) * d.dtype.itemsize))
                for i in range(0, d.shape[0], step):          # through the library's filter pipeline, in row blocks
                    h.update(np.ascontiguousarray(d[i:i + step]).tobytes())
                rec["sha256"] = h.hexdigest()
            out[name] = rec
    return out


def write_station_db(path):
    """What netCDF-C writes for the reference's database: dimensions as HDF5 dimension scales carrying ``_Netcdf4Dimid``,
    coordinate variables AS the scale of their dimension, every other variable attached to its dimensions' scales,
    variable-length UTF-8 string ids, ``_FillValue`` attributes of one element, gzip + shuffle ``(ndays, 1)`` chunks."""
    rng = np.random.default_rng(17)
    n, nd = 23, 59
    ids = sorted("GHCN_USW%08d" % i for i in rng.choice(10 ** 6, n, replace=False))
    lon, lat, elev = rng.uniform(-115, -105, n), rng.uniform(42, 48, n), rng.uniform(500, 3000, n)
    obs = rng.normal(-3, 9, (nd, n)).astype("f4")
    norm = rng.normal(-5, 4, n)
    norm[3] = 9.969209968386869e36                              # a never-written entry: the default fill
    with h5py.File(path, "w", track_order=True) as f:
        f.attrs["_NCProperties"] = np.bytes_("version=2,netcdf=4.7.4,hdf5=1.10.6")
        st = h5py.string_dtype("utf-8")
        t = f.create_dataset("time", data=np.arange(nd, dtype="f8") + 366.0, track_order=True)
        t.attrs["units"] = np.bytes_("days since 1948-1-1 0:0:0")
        t.attrs["calendar"] = np.bytes_("standard")
        t.make_scale("time")
        t.attrs["_Netcdf4Dimid"] = np.int32(0)
        sid = f.create_dataset("station_id", data=np.array(ids, dtype=object), dtype=st, track_order=True)
        sid.make_scale("station_id")
        sid.attrs["_Netcdf4Dimid"] = np.int32(1)

        def col(name, data, fill=None, dtype="f8"):
            d = f.create_dataset(name, data=np.asarray(data, dtype), fillvalue=fill, track_order=True)
            if fill is not None:
                d.attrs["_FillValue"] = np.array([fill], dtype)
                d.attrs["missing_value"] = np.array([fill], dtype)
            d.dims[0].attach_scale(sid)
            return d
        col("longitude", lon, -9999.0)
        col("latitude", lat, -9999.0)
        col("elevation", elev, -9999.0)
        col("tdi", rng.uniform(0, 100, n))
        col("norm01", norm)                                        # no _FillValue attribute: default fill means missing
        nm = f.create_dataset("station_name", data=np.array(["STATION %02d" % i for i in range(n)], dtype=object), dtype=st, track_order=True)
        nm.dims[0].attach_scale(sid)
        ov = f.create_dataset("tmin", data=obs, chunks=(nd, 1), compression="gzip", compression_opts=4, shuffle=True,
                              fillvalue=np.float32(9.969209968386869e36), track_order=True)
        ov.attrs["_FillValue"] = np.array([9.969209968386869e36], "f4")
        ov.attrs["units"] = np.bytes_("C")
        ov.dims[0].attach_scale(t)
        ov.dims[1].attach_scale(sid)
    return {"ids": ids, "lon": lon.tolist(), "lat": lat.tolist(), "elev": elev.tolist(), "norm01": [None if v > 1e36 else v for v in norm.tolist()],
            "obs_sha256": hashlib.sha256(obs.tobytes()).hexdigest(), "ndays": nd, "first_day": 19490101}


if __name__ == "__main__":
    cmd, path = sys.argv[1], sys.argv[2]
    print(json.dumps(describe(path) if cmd == "describe" else write_station_db(path)))
