"""Bandwidth of the monthly aggregation kernel on one year of a 1250 x 1500 daily mosaic (device
resident): algorithmic bytes = 2 B per cell-day read + 2 B per cell-month written."""
import datetime as dt
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from topowx_amd import _lib  # noqa: E402
from topowx_amd.dates import MONTH, YEAR, get_days_metadata  # noqa: E402

days = get_days_metadata(dt.date(2001, 1, 1), dt.date(2001, 12, 31))
ctx = _lib.Context(0)
ctx.set_days(days)
res = {}
for Y, X in ((1250, 1500), (3250, 7000)):
    ncell = Y * X
    nd = days.size if ncell < 4e6 else 120          # the full CONUS year does not fit: 120 days of it
    if nd != days.size:
        d2 = days[:nd]
        ctx.set_days(d2)
    nyr, nmth = ctx.aggregate_dims()
    daily = torch.randint(-3000, 3500, (nd, ncell), dtype=torch.int16, device="cuda")
    out = torch.empty((nyr * nmth, ncell), dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ms = [ctx.aggregate_dev(daily.data_ptr(), 0, ncell, mthly_i16_ptr=out.data_ptr(), stream=s) for _ in range(6)][1:]
    b = nd * ncell * 2 + nyr * nmth * ncell * 2
    res["%dx%d" % (Y, X)] = dict(days=int(nd), ms=float(np.mean(ms)), GBps=b / np.mean(ms) / 1e6,
                                 cell_days_per_s=nd * ncell / np.mean(ms) * 1e3)
    # f8 output as well (daily_to_mthly)
    outf = torch.empty((nyr * nmth, ncell), dtype=torch.float64, device="cuda")
    ms = [ctx.aggregate_dev(daily.data_ptr(), 0, ncell, mthly_ptr=outf.data_ptr(), stream=s) for _ in range(4)][1:]
    res["%dx%d_f8out" % (Y, X)] = dict(ms=float(np.mean(ms)), GBps=(nd * ncell * 2 + nyr * nmth * ncell * 8) / np.mean(ms) / 1e6)
    del daily, out, outf
print(json.dumps(res))
