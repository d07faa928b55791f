#!/bin/bash
# Copy what tests/tools/collect_round.sh left under gpurun_out/ into profiles/ under this round's names.
# usage: bash tests/tools/install_profiles.sh r2
set -eu
R=${1:?round tag, e.g. r2}
P=gpurun_out/prof_round
cp $P/bench.json profiles/${R}_bench.json
cp $P/bench_profiled.json profiles/${R}_bench_profiled.json
cp $P/bench_daily_profiled.json profiles/${R}_bench_daily_profiled.json
cp $P/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $P/kernel_stats_daily.csv profiles/${R}_daily_kernel_stats.csv
cp $P/pmc_FETCH_SIZE.csv profiles/${R}_bench_pmc_FETCH_SIZE.csv
cp $P/pmc_WRITE_SIZE.csv profiles/${R}_bench_pmc_WRITE_SIZE.csv
cp $P/pmc_daily_FETCH_SIZE.csv profiles/${R}_daily_pmc_FETCH_SIZE.csv
cp $P/pmc_daily_WRITE_SIZE.csv profiles/${R}_daily_pmc_WRITE_SIZE.csv
cp $P/hbm_traffic.json profiles/${R}_bench_hbm_traffic.json
cp $P/c4_stream_daily.json profiles/${R}_c4_stream_daily.json
cp $P/xval_c5.json profiles/${R}_xval_c5.json
cp gpurun_out/sq_krig/sq_table.txt profiles/${R}_sq_counters_kriging.txt
cp gpurun_out/sq_daily/sq_table.txt profiles/${R}_sq_counters_daily.txt
cp $P/closepair_scan.json profiles/${R}_closepair_scan.json
cp $P/closepair_scan_fast.json profiles/${R}_closepair_scan_fast_only.json
cp $P/f64_cost.json profiles/${R}_f64_build_cost.json
cp topowx_amd/libtwxhip.resources.txt profiles/${R}_isa_resources.txt
cp gpurun_out/prof_c4/c4_daily_traffic.json profiles/${R}_c4_daily_traffic.json
cp gpurun_out/prof_c4/pmc_daily_FETCH_SIZE.csv profiles/${R}_c4_daily_pmc_FETCH_SIZE.csv
cp gpurun_out/prof_c4/pmc_daily_WRITE_SIZE.csv profiles/${R}_c4_daily_pmc_WRITE_SIZE.csv
[ -f $P/cycle_floor.md ] && cp $P/cycle_floor.md profiles/${R}_cycle_floor.md
[ -f $P/host_page_rates.json ] && cp $P/host_page_rates.json profiles/${R}_host_page_rates.json
[ -f $P/deflate_kernels.json ] && cp $P/deflate_kernels.json profiles/${R}_deflate_kernels.json
