"""GPU: the ctypes stub of INTEGRATION.md section B -- what a maintainer of the reference would paste into
twx/interp/ -- executed VERBATIM (the text of the document's code block; only the library's path is substituted) on
a work chunk of the golden case, and checked against the oracle.  No topowx_amd._lib involved."""
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _section_b_code():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Bind the C ABI directly"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    return m.group(1)


def test_integration_md_section_b_runs_verbatim(golden_case, orc):
    from topowx_amd.dates import MONTH, YEAR
    from topowx_amd.interp import Tiler
    grid, tmin, tmax = golden_case
    code = _section_b_code()
    assert 'C.CDLL("libtwxhip.so")' in code
    code = code.replace('C.CDLL("libtwxhip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "topowx_amd", "libtwxhip.so"))
    # the reference's work chunk: f8[32, Y, X] (tiling.py:205-213), here one 12 x 10 window of the golden grid
    rs, cs = slice(30, 42), slice(50, 60)
    Y, X = 12, 10
    sub = {k: (v[rs, cs] if v.ndim == 2 else v[:, rs, cs]) for k, v in grid.items() if hasattr(v, "ndim") and v.ndim >= 2}
    sub["lat"], sub["lon"] = grid["lat"][rs], grid["lon"][cs]
    sub["mask"] = np.array(sub["mask"], copy=True)
    sub["mask"][0, :3] = 0
    sub["bbox"] = grid["bbox"]
    _, wrk_chk = next(Tiler(sub, Y, X, Y, X))
    nd = tmin.days.size
    fill_i2, fill_f4, fill_i4 = np.int16(-32767), np.float32(9.969209968386869e36), np.int32(-2147483647)
    ns = dict(
        stndaTmin=tmin, stndaTmax=tmax, ndays=nd, month_i32=np.ascontiguousarray(tmin.days[MONTH], np.int32),
        year_i32=np.ascontiguousarray(tmin.days[YEAR], np.int32), wrk_chk=wrk_chk, Y=Y, X=X,
        mask_u8=np.ascontiguousarray(wrk_chk[2] != 0, np.uint8), lat_f8=np.ascontiguousarray(wrk_chk[3, :, 0]),
        lon_f8=np.ascontiguousarray(wrk_chk[4, 0, :]),
        rslt_tmin=np.full((nd, Y, X), fill_i2, np.int16), rslt_tmax=np.full((nd, Y, X), fill_i2, np.int16),
        rslt_tmin_norm=np.full((12, Y, X), fill_f4, np.float32), rslt_tmin_se=np.full((12, Y, X), fill_f4, np.float32),
        rslt_tmax_norm=np.full((12, Y, X), fill_f4, np.float32), rslt_tmax_se=np.full((12, Y, X), fill_f4, np.float32),
        rslt_ninvalid=np.full((Y, X), fill_i4, np.int32), status=np.full((Y, X), 99, np.int32))
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    want = orc.interp_grid(orc.Db(tmin), orc.Db(tmax), orc.params(), sub, daily=True, nthreads=4)
    m = sub["mask"] != 0
    assert np.array_equal(ns["status"], want["status"]) and np.all(ns["status"][m] == 0) and np.all(ns["status"][~m] == -1)
    assert np.array_equal(ns["rslt_ninvalid"], want["ninvalid"])
    for k, w in (("rslt_tmin_norm", "norm_tmin"), ("rslt_tmin_se", "se_tmin"), ("rslt_tmax_norm", "norm_tmax"), ("rslt_tmax_se", "se_tmax")):
        assert np.abs(ns[k].astype(np.float64) - want[w])[:, m].max() < 1e-4
        assert np.all(ns[k][:, ~m] == fill_f4)                        # masked cells keep the fill value
    for k, w in (("rslt_tmin", "daily_tmin"), ("rslt_tmax", "daily_tmax")):
        assert np.abs(ns[k].astype(np.int32) - want[w].astype(np.int32))[:, m].max() <= 1
        assert np.all(ns[k][:, ~m] == fill_i2)
