#!/bin/bash
# Per-kernel TOTAL durations of one bench.py config record (rocprofv3 --kernel-trace --stats), e.g. c3 against c3_fitted:
#   gpurun -- bash tests/tools/config_stats.sh c3_fitted [tag]
set -u
CFG=${1:-c3_fitted}; TAG=${2:-cs_$CFG}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-daily --configs $CFG --c3-steps 1 2>$OUT/stats.err | tail -1 > $OUT/bench_profiled.json
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, json, os, sys
out, cfg = sys.argv[1], sys.argv[2]
p = glob.glob(os.path.join(out, "stats", "**", "s_kernel_stats.csv"), recursive=True)
rows = list(csv.DictReader(open(p[0])))
lines = []
for r in rows:
    k = r["Name"].split("(")[0].replace("void ", "")
    if k.startswith(("k_", "__amd")):
        lines.append((float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, k))
with open(os.path.join(out, "kernel_totals.txt"), "w") as fh:
    for ms, n, us, k in sorted(lines, reverse=True):
        s = "%10.2f ms total  %7d x %9.1f us  %s" % (ms, n, us, k)
        print(s); fh.write(s + "\n")
    s = "sum %.1f ms" % sum(l[0] for l in lines)
    print(s); fh.write(s + "\n")
try:
    d = json.loads(open(os.path.join(out, "bench_profiled.json")).read())
    c = d["configs"][cfg]
    print(cfg, "%.4g %s, %.1f ms per step" % (c["value"], c["unit"].split(" ")[0], c["ms_per_step"]))
except Exception as e:
    print("bench line unreadable:", e)
PY
