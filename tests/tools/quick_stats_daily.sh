#!/bin/bash
# Per-kernel average durations of the daily record (10-year tile) under rocprofv3:  gpurun -- bash tests/tools/quick_stats_daily.sh [tag]
set -u
TAG=${1:-qsd}; shift || true
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-configs --stream-tiles 0 "$@" 2>$OUT/stats.err | tail -1 > $OUT/bench_profiled.json
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
p = glob.glob(os.path.join(out, "stats", "**", "s_kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(p[0])):
    k = r["Name"].split("(")[0].replace("void ", "")
    if k.startswith(("k_daily", "k_tile_u", "k_perm", "k_gwr", "k_fix", "k_row", "k_compact")):
        print("%10.1f us x %3d  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), k))
d = json.loads(open(os.path.join(out, "bench_profiled.json")).read())
print("daily: %.4g cell-days/s, %.2f ms per step" % (d["daily"]["value"], d["daily"]["ms_per_step"]), d["daily"]["timing_ms"])
PY
