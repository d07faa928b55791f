#!/bin/bash
# Same-box A/B of library variants on the fp64-covariance-build cost (tests/tools/gpu_f64_cost.py: C2 tile, every system routed):
#   gpurun -- bash tests/tools/ab_f64.sh expold expnew ...     (variants: ab/libtwxhip_NAME.so, tests/tools/build_variant.sh)
# The LAST variant named stays installed as topowx_amd/libtwxhip.so.
set -u
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  python3 tests/tools/gpu_f64_cost.py 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v  f64 %.3f ms  fast %.3f ms  ratio %.3f  maxdiff %.2e' % (d['fp64_build']['uk_ms'], d['fast_only']['uk_ms'], d['ratio'], d['fast_vs_fp64_max_abs_degC']))" | tee -a gpurun_out/ab_f64.txt
done
done
