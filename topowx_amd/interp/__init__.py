"""Python-3 facade with the class / method names of ``twx.interp``.

Every numerical result comes from libtwxhip (HIP kernels on MI355X) through
``topowx_amd._lib``; there is no CPU implementation here.  ``__all__`` holds every name of the reference's four
``__all__`` lists (interp_tair.py:22-24, station_select.py:23, optimize.py:20-23, tiling.py:23) except ``XvalOutlier``
(station QA upstream of the serially-complete database: out of scope).
"""
from .station_select import StationSelect  # noqa: F401
from .interp_tair import (BuildKrigParams, GwrTairAnom, InterpTair, KrigTair, KrigTairAll,  # noqa: F401
                          PredictorGrids, PtInterpTair, StationDataWrkChk, build_empty_pt, tmin_tmax_fixer)
from .optimize import (StationKrigParams, XvalTairAnom, XvalTairNorm, XvalTairOverall,  # noqa: F401
                       build_nstn_bandwidths)
from .tiling import Tiler, TileGridInfo  # noqa: F401
from .aggregate import TairAggregate, TileMosaic, mthly_from_daily, write_ds_mthly  # noqa: F401
from ..ncio import TileWriter, create_climdiv_optim_nstns_db  # noqa: F401


def set_optim_nstns_tair_norm(stnda, path_xval_ds, *args, **kwargs):
    """optimize.py:268-316: ``(stnda, path_xval_ds)`` as in the reference, or the array form of ``topowx_amd.xval``."""
    from ..xval import set_optim_nstns_tair_norm as f
    return f(stnda, path_xval_ds, *args, **kwargs)


def set_optim_nstns_tair_anom(stnda, path_xval_ds, *args, **kwargs):
    """optimize.py:318-374."""
    from ..xval import set_optim_nstns_tair_anom as f
    return f(stnda, path_xval_ds, *args, **kwargs)


__all__ = ["StationSelect", "KrigTair", "KrigTairAll", "BuildKrigParams", "GwrTairAnom", "InterpTair",
           "PtInterpTair", "PredictorGrids", "StationDataWrkChk", "build_empty_pt", "tmin_tmax_fixer", "XvalTairOverall",
           "XvalTairAnom", "XvalTairNorm", "StationKrigParams", "build_nstn_bandwidths", "Tiler", "TileGridInfo",
           "TileWriter", "TairAggregate", "TileMosaic", "mthly_from_daily", "write_ds_mthly",
           "create_climdiv_optim_nstns_db", "set_optim_nstns_tair_norm", "set_optim_nstns_tair_anom"]
