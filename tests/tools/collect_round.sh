#!/bin/bash
# One call that refreshes everything kept under profiles/ for a round: headline + daily profiles (collect_profiles.sh),
# the C4 daily / streamed record, the SQ counter tables and the config-5 cross-validation timings.
# gpurun -- bash tests/tools/collect_round.sh
set -u
mkdir -p gpurun_out
# the C4 tile's counter passes first (bench.py's configs.c4_tile.traffic quotes the newest profiles/r*_c4_daily_traffic.json)
bash tests/tools/collect_c4_traffic.sh > gpurun_out/collect_c4.log 2>&1
NEWEST=$(ls profiles/r*_c4_daily_traffic.json 2>/dev/null | sort | tail -1)
[ -n "${TWX_ROUND_TAG:-}" ] && NEWEST=profiles/${TWX_ROUND_TAG}_c4_daily_traffic.json      # (see collect_profiles.sh)
[ -n "$NEWEST" ] && cp gpurun_out/prof_c4/c4_daily_traffic.json $NEWEST
bash tests/tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1
python3 bench.py --steps 6 --warmup 2 --daily-years 69 --daily-year0 1948 --stream-tiles 4 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/prof_round/c4_stream_daily.json
bash tests/tools/collect_sq.sh sq_krig > /dev/null 2>&1
TWX_SQ_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-configs --stream-tiles 0" bash tests/tools/collect_sq.sh sq_daily > /dev/null 2>&1
python3 -m topowx_amd.xval --db c5 --nstns 12000 --years 3 > gpurun_out/prof_round/xval_c5.json 2> gpurun_out/xval.err
ls gpurun_out/prof_round gpurun_out/sq_krig gpurun_out/sq_daily
# round 4: ill-conditioned kriging systems -- error of the routed and of the fast-only build per (nugget, psill, range), and the
# cost of the fp64 covariance build when EVERY system of the C2 tile takes it
python3 tests/tools/gpu_closepair_scan.py > gpurun_out/prof_round/closepair_scan.log 2>&1
python3 tests/tools/gpu_closepair_scan.py --fast-only > gpurun_out/prof_round/closepair_scan_fast.log 2>&1
python3 tests/tools/gpu_f64_cost.py > gpurun_out/prof_round/f64_cost.log 2>&1
cp gpurun_out/closepair_scan.json gpurun_out/closepair_scan_fast.json gpurun_out/f64_cost.json gpurun_out/prof_round/
# round 6: the cycle floor of the kriging kernels' design and what the host gives a writer of new file pages
python3 tests/tools/cycle_floor.py gpurun_out/sq_krig gpurun_out/prof_round/kernel_stats.csv > gpurun_out/prof_round/cycle_floor.md 2> gpurun_out/prof_round/cycle_floor.err
python3 tests/tools/host_page_rates.py /dev/shm 8 > gpurun_out/prof_round/host_page_rates.log 2>&1
cp gpurun_out/host_page_rates.json gpurun_out/prof_round/
# round 6: the deflate kernels (csrc/twx_deflate.h) on one configs[3]-sized tile: kernel summary, HBM bytes per launch
D=gpurun_out/prof_round/dfl
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D/stats -o s --output-format csv -- python3 tests/tools/gpu_deflate_ab.py --child > $D/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o f --output-format csv -- python3 tests/tools/gpu_deflate_ab.py --child > $D/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o w --output-format csv -- python3 tests/tools/gpu_deflate_ab.py --child > $D/write.log 2>&1
python3 tests/tools/reduce_deflate.py $D > $D/reduce.log 2>&1
cp $D/deflate_kernels.json gpurun_out/prof_round/
