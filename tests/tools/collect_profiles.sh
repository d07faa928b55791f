#!/bin/bash
# Collect the round's bench line, the rocprofv3 kernel summary and the two HBM-traffic PMC passes of bench.py
# (run on the GPU box from the repo root: `gpurun -- bash tests/tools/collect_profiles.sh`); everything lands in
# gpurun_out/prof_round/ and is reduced by tests/tools/reduce_profiles.py into the files kept under profiles/.
# The PMC passes carry --kernel-trace only (no other trace domain), one counter per pass.
set -u
OUT=gpurun_out/prof_round
mkdir -p $OUT
export TMPDIR=/tmp
# (1) the counter passes first: the bench lines below quote roofline.traffic / daily.traffic from the newest
#     profiles/r*_bench_hbm_traffic.json, which therefore has to belong to THIS tree's kernels before they run
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-daily --no-configs > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-daily --no-configs > $OUT/write.log 2>&1
# daily record: kernels that only the daily path launches (k_daily_tile, k_tile_uidx, k_gwr_z_cell, k_fix_sparse, ...)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/dfetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-configs --stream-tiles 0 > $OUT/dfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/dwrite -o w --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-configs --stream-tiles 0 > $OUT/dwrite.log 2>&1
python3 tests/tools/reduce_profiles.py $OUT > /dev/null
# (on the box's copy of the tree; install_profiles.sh does it at home.  TWX_ROUND_TAG=r5 names the file the bench lines will cite;
# without it the newest committed file is overwritten and cited under its old name)
NEWEST=$(ls profiles/r*_bench_hbm_traffic.json 2>/dev/null | sort | tail -1)
[ -n "${TWX_ROUND_TAG:-}" ] && NEWEST=profiles/${TWX_ROUND_TAG}_bench_hbm_traffic.json
[ -n "$NEWEST" ] && cp $OUT/hbm_traffic.json $NEWEST
# (2) the bench line and the kernel summaries
python3 bench.py --steps 20 --warmup 3 2>$OUT/bench.err | tail -1 > $OUT/bench.json
# headline workload alone (the daily record would mix other batch shapes into the same kernels' averages)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-daily --no-configs 2>$OUT/stats.err | tail -1 > $OUT/bench_profiled.json
rocprofv3 --kernel-trace --stats -d $OUT/dstats -o s --output-format csv -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-configs --stream-tiles 0 2>$OUT/dstats.err | tail -1 > $OUT/bench_daily_profiled.json
python3 tests/tools/reduce_profiles.py $OUT
ls $OUT
