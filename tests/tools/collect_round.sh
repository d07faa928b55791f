#!/bin/bash
# One call that refreshes everything kept under profiles/ for a round: headline + daily profiles (collect_profiles.sh),
# the C4 daily / streamed record, the SQ counter tables and the config-5 cross-validation timings.
# gpurun -- bash tests/tools/collect_round.sh
set -u
bash tests/tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1
python3 bench.py --steps 6 --warmup 2 --daily-years 69 --stream-tiles 4 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/prof_round/c4_stream_daily.json
bash tests/tools/collect_sq.sh sq_krig > /dev/null 2>&1
TWX_SQ_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-configs --stream-tiles 0" bash tests/tools/collect_sq.sh sq_daily > /dev/null 2>&1
python3 -m topowx_amd.xval --nstns 10000 --years 3 > gpurun_out/prof_round/xval_10k.json 2> gpurun_out/xval.err
ls gpurun_out/prof_round gpurun_out/sq_krig gpurun_out/sq_daily
