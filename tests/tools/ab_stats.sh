#!/bin/bash
# Same-box A/B of two builds of the library: ab/libtwxhip_old.so vs ab/libtwxhip_new.so (boxes differ by 3-5 % between
# gpurun calls, so two builds are only comparable inside one call):  gpurun -- bash tests/tools/ab_stats.sh [bench args]
set -u
for v in old new old new; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  echo "== $v"
  bash tests/tools/quick_stats.sh ab_$v "$@" | grep "k_uk\|k_gwr\|k_select<\|k_tile_dist\|kriging\|bench"
done
cp ab/libtwxhip_new.so topowx_amd/libtwxhip.so
