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
    """Progress printer (twx/utils/status_check.py:22-57): ``increment()`` reports every ``check_cnt`` items."""

    def __init__(self, total_cnt, check_cnt, out=None):
        self.total_cnt, self.check_cnt = total_cnt, check_cnt
        self.num = self.num_last_check = 0
        self.status_time = self.start_time = time.time()
        self.out = out

    def increment(self, n=1):
        self.num += n
        if self.num - self.num_last_check < self.check_cnt:
            return
        now = time.time()
        out = self.out or sys.stdout
        last = self.num - self.num_last_check
        if self.total_cnt != -1:
            out.write("Total items processed is %d.  Last %d items took %f minutes. %d items to go.\n" % (
                self.num, last, (now - self.status_time) / 60.0, self.total_cnt - self.num))
            out.write("Current total process time: %f minutes\n" % ((now - self.start_time) / 60.0))
            out.write("Estimated Time Remaining: %f\n" % (
                ((self.total_cnt - self.num) / float(self.num)) * ((now - self.start_time) / 60.0)))
        else:
            out.write("Total items processed is %d.  Last %d items took %f minutes\n" % (
                self.num, last, (now - self.status_time) / 60.0))
            out.write("Current total process time: %f minutes\n" % ((now - self.start_time) / 60.0))
        out.flush()
        self.status_time, self.num_last_check = time.time(), self.num


class Unbuffered(object):
    """twx/utils/util_misc.py:26-33: a stream that flushes on every write."""

    def __init__(self, stream):
        self.stream = stream

    def write(self, data):
        self.stream.write(data)
        self.stream.flush()

    def __getattr__(self, attr):
        return getattr(self.stream, attr)


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
