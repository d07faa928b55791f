"""``twx.utils``: what the interpolation-side scripts import (step21..27)."""
from topowx_amd.utils import StatusCheck, TwxConfig, Unbuffered, mkdir_p  # noqa: F401
from topowx_amd.dates import (DATE, DAY, MONTH, YEAR, YMD, get_days_metadata, get_mth_metadata)  # noqa: F401

__all__ = ["StatusCheck", "Unbuffered", "TwxConfig", "mkdir_p", "get_days_metadata", "get_mth_metadata", "DATE", "YMD", "YEAR",
           "MONTH", "DAY"]
