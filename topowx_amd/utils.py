"""Small host utilities the reference's step scripts import from ``twx.utils`` (``scripts/step21..27``: ``StatusCheck``,
``Unbuffered``, ``TwxConfig``, ``mkdir_p``) -- kept so that a py3 translation of those scripts keeps its import lines.
Nothing here is on the hot path."""
import configparser
import datetime as _dt
import os
import sys
import time

__all__ = ["StatusCheck", "Unbuffered", "TwxConfig", "mkdir_p"]


class StatusCheck(object):
    """Progress reporter with the constructor and ``increment`` of the reference's helper (``StatusCheck(total_cnt, check_cnt)``
    in scripts/step21..24): one line every ``check_cnt`` items -- items done, share of the total, rate of the last batch,
    elapsed time and, when the total is known (``total_cnt`` != -1), items left and an estimate of the time left.
    Own wording and bookkeeping; nothing parses these lines."""

    def __init__(self, total_cnt, check_cnt, out=None):
        self.total_cnt, self.check_cnt, self.out = total_cnt, max(int(check_cnt), 1), out
        self.num = 0
        self._reported = 0
        self._t_start = self._t_batch = time.monotonic()

    def increment(self, n=1):
        self.num += n
        batch = self.num - self._reported
        if batch < self.check_cnt:
            return
        now = time.monotonic()
        elapsed, took = (now - self._t_start) / 60.0, (now - self._t_batch) / 60.0
        parts = ["[progress] %d done" % self.num]
        if self.total_cnt != -1:
            left = self.total_cnt - self.num
            parts[0] += " of %d (%.1f %%)" % (self.total_cnt, 100.0 * self.num / max(self.total_cnt, 1))
            parts.append("%d left" % left)
            parts.append("about %.2f min to go" % (elapsed * left / float(self.num)))
        parts.insert(1, "last %d in %.2f min" % (batch, took))
        parts.append("%.2f min so far" % elapsed)
        stream = self.out if self.out is not None else sys.stdout
        stream.write(", ".join(parts) + "\n")
        stream.flush()
        self._reported, self._t_batch = self.num, time.monotonic()


class Unbuffered(object):
    """``sys.stdout = Unbuffered(sys.stdout)`` of the step scripts: every ``write`` / ``writelines`` reaches the wrapped stream
    at once; everything else is the stream's own."""
    __slots__ = ("_wrapped",)

    def __init__(self, stream):
        object.__setattr__(self, "_wrapped", stream)

    def write(self, text):
        n = self._wrapped.write(text)
        self._wrapped.flush()
        return n

    def writelines(self, lines):
        self._wrapped.writelines(lines)
        self._wrapped.flush()

    def __getattr__(self, name):
        return getattr(self._wrapped, name)


def mkdir_p(path):
    os.makedirs(path, exist_ok=True)


def _ymd(s):
    d = _dt.datetime.strptime(s.strip(), "%Y-%m-%d")
    return d.year * 10000 + d.month * 100 + d.day


class TwxConfig(object):
    """The INI file of a TopoWx run (twx/utils/config.py:7-205), reduced to what the interpolation steps read: the data
    root, the interpolation period, the dataset version, and the directory layout under the root --
    ``fpath_stndata_nc_serial_tmin / _tmax`` (step21-25), ``path_interp_optim_norms / _anoms`` (step21, step23),
    ``fpath_xval_interp_nc_tmin / _tmax`` (step24), ``path_predictor_rasters``, ``path_tile_out`` (step25),
    ``path_mosaic_norms / _daily / _monthly`` (step26, step27).  Directories are created as the reference creates them."""

    def __init__(self, fpath_ini):
        cfg = configparser.ConfigParser()
        if not cfg.read(fpath_ini):
            raise IOError("cannot read TopoWx configuration: %s" % fpath_ini)
        sec = cfg["TOPOWX_CONFIG"]
        self.twx_data_root = sec["twx_data_root"]
        self.interp_start_date = _dt.datetime.strptime(sec.get("interp_start_date", "1948-01-01"), "%Y-%m-%d")
        self.interp_end_date = _dt.datetime.strptime(sec.get("interp_end_date", "2016-12-31"), "%Y-%m-%d")
        self.twx_data_version = sec.get("twx_data_version", "1.0.0")
        if "stn_bbox" in sec:
            self.stn_bbox = tuple(float(x) for x in sec["stn_bbox"].split(","))
        j = os.path.join
        self.path_stndata = j(self.twx_data_root, "station_data")
        self.path_stndata_infill = j(self.path_stndata, "infill")
        self.fpath_stndata_nc_serial_tmin = j(self.path_stndata_infill, "serial_tmin.nc")
        self.fpath_stndata_nc_serial_tmax = j(self.path_stndata_infill, "serial_tmax.nc")
        self.path_interp_optim_norms = j(self.path_stndata_infill, "optim_norm")
        self.path_interp_optim_anoms = j(self.path_stndata_infill, "optim_anom")
        self.fpath_xval_interp_nc_tmin = j(self.path_stndata_infill, "xval_interp_tmin.nc")
        self.fpath_xval_interp_nc_tmax = j(self.path_stndata_infill, "xval_interp_tmax.nc")
        self.path_rasters = j(self.twx_data_root, "rasters")
        self.path_predictor_rasters = j(self.path_rasters, "conus_interp_grids", "ncdf")
        self.path_tile_out = j(self.twx_data_root, "tile_output")
        self.path_logs = j(self.twx_data_root, "logs")
        self.path_final_output = j(self.twx_data_root, "final_output_data")
        self.path_mosaic_norms = j(self.path_final_output, "normals")
        self.path_mosaic_daily = j(self.path_final_output, "daily")
        self.path_mosaic_monthly = j(self.path_final_output, "monthly")
        for p in (self.path_stndata_infill, self.path_interp_optim_norms, self.path_interp_optim_anoms,
                  self.path_predictor_rasters, self.path_tile_out, self.path_logs, self.path_mosaic_norms,
                  self.path_mosaic_daily, self.path_mosaic_monthly):
            mkdir_p(p)
