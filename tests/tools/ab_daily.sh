#!/bin/bash
# Same-box A/B of two builds on the C4 tile (25 203 days) daily record: ab/libtwxhip_old.so vs ab/libtwxhip_new.so
#   gpurun -- bash tests/tools/ab_daily.sh
set -u
for v in old new old new; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  echo "== $v"
  python3 bench.py --steps 4 --warmup 1 --daily-years 69 --stream-tiles 0 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['daily']['timing_ms']; print('C4 tile %.2f ms  daily %.2f gwr %.2f fix %.2f uk %.2f select %.2f' % (d['daily']['ms_per_step'], t['daily_ms'], t['gwr_ms'], t['fix_ms'], t['uk_ms'], t['select_ms']))"
done
cp ab/libtwxhip_new.so topowx_amd/libtwxhip.so
