"""``twx.db``: the station database class and the field-name constants / helpers of twx/db/station_data.py:22-124."""
from topowx_amd.stationdb import (BAD, CLIMDIV, ELEV, LAT, LON, LST, MASK, NORM, OPTIM_NNGH, OPTIM_NNGH_ANOM, STN_ID, TDI,  # noqa: F401
                                  VARIO_NUG, VARIO_PSILL, VARIO_RNG, StationSerialDataDb, get_krigparam_varname,
                                  get_lst_varname, get_norm_varname, get_optim_anom_varname, get_optim_varname)
from topowx_amd.dates import DATE, DAY, MONTH, YEAR, YMD  # noqa: F401
from topowx_amd.ncio import create_quick_db  # noqa: F401

STN_NAME = "station_name"
STATE = "state"

__all__ = ["LON", "LAT", "ELEV", "STN_ID", "STN_NAME", "STATE", "StationSerialDataDb", "BAD", "CLIMDIV", "MASK", "TDI",
           "get_norm_varname", "get_optim_varname", "get_optim_anom_varname", "get_lst_varname", "get_krigparam_varname",
           "VARIO_NUG", "VARIO_PSILL", "VARIO_RNG", "LST", "NORM", "OPTIM_NNGH", "OPTIM_NNGH_ANOM", "DATE", "YMD", "YEAR",
           "MONTH", "DAY", "create_quick_db"]
