"""Device time of the deflate kernels (twx_deflate.h) on one configs[3]-sized tile (250 x 250 cells x 25 203 days, chunks of
50 x 50), per library variant -- a same-box A/B (variants: ab/libtwxhip_NAME.so, tests/tools/build_variant.sh):
    python3 tests/tools/gpu_deflate_ab.py NAME [NAME ...]      (the installed library is restored afterwards)
    python3 tests/tools/gpu_deflate_ab.py                      (the installed library only)
Few stations (the setup generates 69 years of observations per station); the deflate kernels do not care."""
import datetime as dt
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "topowx_amd", "libtwxhip.so")


def child():
    import numpy as np
    from topowx_amd import _lib, synth
    from topowx_amd.dates import get_days_metadata
    days = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
    grid = synth.make_grid("C2", nrows=250, ncols=250)
    tmin = synth.make_stations(grid["bbox"], 400, 1, "tmin", days, with_obs=True)
    tmax = synth.make_stations(grid["bbox"], 400, 1, "tmax", days, with_obs=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    st = ctx.stream(250, 250, daily=True, nslots=2, deflate_chunks=(50, 50))
    ms, tot = [], 0
    for rep in range(5):
        st.submit(rep & 1, grid)
        o = st.wait(rep & 1)
        tot = sum(len(b) for v in ("tmin", "tmax") for b in o["deflated_" + v])
        ms.append(ctx.timing()["deflate_ms"])
    st.close()
    ctx.close()
    print(json.dumps({"deflate_ms": [round(m, 2) for m in ms], "median": float(np.median(ms[1:])), "bytes_over_int16": tot / (2 * 2 * days.size * 62500)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    names = sys.argv[1:]
    keep = LIB + ".keep"
    if names:
        shutil.copy(LIB, keep)
    try:
        for rep in range(2 if names else 1):
            for n in names or ["installed"]:
                if names:
                    shutil.copy(os.path.join(ROOT, "ab", "libtwxhip_%s.so" % n), LIB)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True)
                line = [ln for ln in r.stdout.split("\n") if ln.startswith("{")]
                print("SUMMARY", n, line[-1] if line else ("failed: " + r.stderr[-300:]), flush=True)
    finally:
        if names:
            shutil.move(keep, LIB)
