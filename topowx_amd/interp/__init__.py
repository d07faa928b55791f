"""Python-3 facade with the class / method names of ``twx.interp``.

Every numerical result comes from libtwxhip (HIP kernels on MI355X) through
``topowx_amd._lib``; there is no CPU implementation here.
"""
from .station_select import StationSelect  # noqa: F401
from .interp_tair import (BuildKrigParams, GwrTairAnom, InterpTair, KrigTair, KrigTairAll,  # noqa: F401
                          PredictorGrids, PtInterpTair, StationDataWrkChk, build_empty_pt, tmin_tmax_fixer)
from .optimize import (StationKrigParams, XvalTairAnom, XvalTairNorm, XvalTairOverall,  # noqa: F401
                       build_nstn_bandwidths)
from .tiling import Tiler, TileGridInfo  # noqa: F401
from .aggregate import TairAggregate, TileMosaic, mthly_from_daily  # noqa: F401

__all__ = ["StationSelect", "KrigTair", "KrigTairAll", "BuildKrigParams", "GwrTairAnom", "InterpTair",
           "PtInterpTair", "PredictorGrids", "StationDataWrkChk", "build_empty_pt", "tmin_tmax_fixer", "XvalTairOverall",
           "XvalTairAnom", "XvalTairNorm", "StationKrigParams", "build_nstn_bandwidths", "Tiler", "TileGridInfo",
           "TairAggregate", "TileMosaic", "mthly_from_daily"]
