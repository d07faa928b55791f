"""Seeded random tiles through a deflate stream (twx_stream_deflate), GPU against zlib and against the CPU restatement:
random tile shapes (4..48 cells a side), chunk shapes that divide them (odd widths included: the one-value-per-load kernels),
random mask holes, one or both variables, the golden case's 1 096 days.  Per case: every chunk's stream inflated by zlib equals
the shuffled chunk of the synchronous entry's values; the first and the last chunk's bytes equal oracle/deflate_oracle.py's.
    python3 tests/tools/gpu_soak_deflate.py [cases] [first seed]   ->  gpurun_out/soak_deflate.json"""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402

import make_golden  # noqa: E402
from oracle import deflate_oracle as dorc  # noqa: E402
from topowx_amd import _lib  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
grid0, tmin, tmax = make_golden.case_inputs()
nd = tmin.days.size
ctx = _lib.Context()
ctx.set_stations(_lib.TMIN, tmin)
ctx.set_stations(_lib.TMAX, tmax)
Yg, Xg = grid0["mask"].shape
res = {"cases": 0, "failed": [], "chunks": 0, "bytes_over_int16": [], "odd_chunk_width": 0, "single_variable": 0, "stored_only_tiles": 0}
t0 = time.time()
for seed in range(seed0, seed0 + ncases):
    rng = np.random.default_rng(1000 + seed)
    cy, cx = int(rng.integers(1, 13)), int(rng.integers(1, 13))
    Y, X = cy * int(rng.integers(1, max(2, 48 // cy))), cx * int(rng.integers(1, max(2, 48 // cx)))
    r0, c0 = int(rng.integers(0, Yg - Y + 1)), int(rng.integers(0, Xg - X + 1))
    variables = ("tmin", "tmax") if rng.random() < 0.7 else (("tmin",) if rng.random() < 0.5 else ("tmax",))
    grid = dict(grid0)
    mask = np.array(grid0["mask"], copy=True)
    if rng.random() < 0.5:
        hole = rng.random((Yg, Xg)) < rng.random() * 0.6
        mask[hole] = 0
    grid["mask"] = mask
    rows, cols = slice(r0, r0 + Y), slice(c0, c0 + X)
    try:
        want = ctx.interp_grid(grid, variables=variables, daily=True, rows=rows, cols=cols)
        st = ctx.stream(Y, X, variables=variables, daily=True, nslots=2, deflate_chunks=(cy, cx))
        st.submit(0, grid, rows, cols)
        o = st.wait(0)
        got = {k: [bytes(b) for b in v] for k, v in o.items() if k.startswith("deflated_")}
        st.close()
        tot = 0
        for var in variables:
            chunks = dorc._chunks(want["daily_" + var], cy, cx)
            blobs = got["deflated_" + var]
            assert len(blobs) == len(chunks), "chunk count"
            for blob, chunk in zip(blobs, chunks):
                lo, hi = dorc.shuffled(chunk)
                assert zlib.decompress(blob) == lo.tobytes() + hi.tobytes(), "inflate"
                tot += len(blob)
            table = dorc.tile_table(want["daily_" + var], cy, cx)
            for c in {0, len(chunks) - 1}:
                assert blobs[c] == dorc.deflate_chunk(chunks[c], table), "bytes differ from the restatement (chunk %d)" % c
            res["chunks"] += len(chunks)
        assert ("deflated_tmin" in got) == ("tmin" in variables) and ("deflated_tmax" in got) == ("tmax" in variables)
        res["bytes_over_int16"].append(tot / (len(variables) * nd * Y * X * 2))
        res["odd_chunk_width"] += cx % 2
        res["single_variable"] += len(variables) == 1
    except Exception as e:                                        # noqa: BLE001 -- recorded, the soak goes on
        res["failed"].append({"seed": seed, "Y": Y, "X": X, "cy": cy, "cx": cx, "variables": variables, "error": repr(e)[:300]})
    res["cases"] += 1
ctx.close()
r = res.pop("bytes_over_int16")
res["bytes_over_int16_min_median_max"] = [float(np.min(r)), float(np.median(r)), float(np.max(r))] if r else None
res["seconds"] = round(time.time() - t0, 1)
res["seeds"] = [seed0, seed0 + ncases - 1]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "soak_deflate.json"), "w"), indent=1)
print("SUMMARY", json.dumps(res))
