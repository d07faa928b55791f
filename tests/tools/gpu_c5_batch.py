"""step21 (config 5) wall time against the station batch size of xval.optim_nstns_norms on the 12 000-station seed-2 database:
    python tests/tools/gpu_c5_batch.py        (on the GPU box; what picked the default batch)"""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from topowx_amd import synth, xval  # noqa: E402
from topowx_amd.dates import get_days_metadata  # noqa: E402

days = get_days_metadata(dt.date(1981, 1, 1), dt.date(1983, 12, 31))
bbox, seed = xval.config5_bbox("c5")
stn = synth.make_stations(bbox, 12000, seed, "tmin", days, with_obs=True)
ids = xval.xval_station_ids(stn)
for b in (256, 512, 1024, 2048, 256, 1024):
    t0 = time.perf_counter()
    xval.optim_nstns_norms(stn, "tmin", stn_ids=ids, batch=b)
    print("batch %5d  step21 %.3f s" % (b, time.perf_counter() - t0), flush=True)
