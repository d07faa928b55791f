#!/bin/bash
# HBM traffic of the daily launches on the FULL 1948-2016 tile (configs[3]): the hat rows are per (cell, month), not per day,
# so their share of the traffic shrinks with the length of the day axis.  gpurun -- bash tests/tools/collect_c4_traffic.sh
set -u
OUT=gpurun_out/prof_c4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --daily-years 69 --no-cpu-baseline --no-configs --stream-tiles 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/dfetch -o f --output-format csv -- python3 bench.py $ARGS > $OUT/dfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/dwrite -o w --output-format csv -- python3 bench.py $ARGS > $OUT/dwrite.log 2>&1
python3 tests/tools/reduce_profiles.py $OUT > /dev/null
python3 - <<'PY'
import json
t = json.load(open("gpurun_out/prof_c4/hbm_traffic.json"))
f, w = t["FETCH_SIZE"]["daily_path_per_step_bytes"], t["WRITE_SIZE"]["daily_path_per_step_bytes"]
alg = 2.03 * 62500 * 25203 * 2
out = {"workload": "configs[3] tile: 250x250 cells, 25 203 days, Tmin + Tmax", "fetched_bytes_per_step": f, "written_bytes_per_step": w,
       "algorithmic_bytes_per_step": alg, "ratio": (f + w) / alg, "per_kernel": {k: {"fetch": t["FETCH_SIZE"][k]["per_launch_bytes"], "write": t["WRITE_SIZE"][k]["per_launch_bytes"], "launches_per_step": t["FETCH_SIZE"][k]["launches"]}
       for k in ("k_daily_tile", "k_gwr_z_cell", "k_tile_uidx", "k_perm", "k_fix_cells", "k_fix_sparse")}, "kernel_sources_sha16": t["kernel_sources_sha16"]}
json.dump(out, open("gpurun_out/prof_c4/c4_daily_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
