#!/bin/bash
# Same-box A/B of two builds on the headline workload WITHOUT the profiler (bench.py's own HIP-event timings):
# ab/libtwxhip_old.so vs ab/libtwxhip_new.so.   gpurun -- bash tests/tools/ab_bench.sh
set -u
for v in old new old new old new; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-daily --no-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['timing_ms']; print('$v  %.3e cell-months/s  step %.3f ms  uk %.3f  select %.3f  parity %.2e' % (d['value'], d['ms_per_step'], t['uk_ms'], t['select_ms'], d.get('parity_max_abs_degC', -1)))"
done
cp ab/libtwxhip_new.so topowx_amd/libtwxhip.so
