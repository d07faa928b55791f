#!/bin/bash
# Per-kernel average durations of the headline workload (rocprofv3 --kernel-trace --stats), printed as a table:
#   gpurun -- bash tests/tools/quick_stats.sh [tag] [extra bench.py args]
set -u
TAG=${1:-qs}; shift || true
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-daily --no-configs "$@" 2>$OUT/stats.err | tail -1 > $OUT/bench_profiled.json
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
p = glob.glob(os.path.join(out, "stats", "**", "s_kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(p[0])))
tot = 0.0
lines = []
for r in rows:
    k = r["Name"].split("(")[0].replace("void ", "")
    if k.startswith(("k_", "__amd")):
        lines.append((float(r["AverageNs"]) / 1e3, int(r["Calls"]), k))
for us, n, k in sorted(lines, reverse=True):
    print("%10.1f us x %3d  %s" % (us, n, k))
krig = sum(us for us, n, k in lines if k.startswith(("k_uk<", "k_ukz<", "k_ukw<", "k_ukwz<", "k_ukw2<", "k_tile_dist", "k_cell_dist")))
print("kriging kernels (sum of averages): %.3f ms" % (krig / 1e3))
try:
    d = json.loads(open(os.path.join(out, "bench_profiled.json")).read())
    print("bench: %.4g cell-months/s, %.3f ms per step, uk_ms %.3f" % (d["value"], d["ms_per_step"], d["timing_ms"]["uk_ms"]))
except Exception as e:
    print("bench line unreadable:", e)
with open(os.path.join(out, "kernel_table.txt"), "w") as fh:
    for us, n, k in sorted(lines, reverse=True):
        fh.write("%10.1f us x %3d  %s\n" % (us, n, k))
PY
