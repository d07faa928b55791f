#!/bin/bash
# Same-box A/B of library variants on the full configs[2] grid, plain and under fitted variograms (bench.py configs c3, c3_fitted):
#   gpurun -- bash tests/tools/ab_c3.sh coarse side ...      (variants: ab/libtwxhip_NAME.so; the LAST one stays installed)
set -u
mkdir -p gpurun_out
for v in "$@"; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  python3 bench.py --no-daily --no-cpu-baseline --configs c3,c3_fitted 2>/dev/null | tail -1 > gpurun_out/ab_c3_$v.json
  python3 -c "
import json
c = json.load(open('gpurun_out/ab_c3_$v.json'))['configs']
a, b = c['c3']['ms_per_step'], c['c3_fitted']['ms_per_step']
print('$v  c3 %.1f ms  c3_fitted %.1f ms  ratio %.4f  fp64 frac %.4f' % (a, b, b / a, c['c3_fitted']['frac_on_fp64_covariance_build']))" | tee -a gpurun_out/ab_c3.txt
done
