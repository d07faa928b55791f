#!/bin/bash
# Do two builds of the library give the same BITS?  ab/libtwxhip_old.so vs ab/libtwxhip_new.so on tests/tools/gpu_out_hash.py
# (a restructured kernel that performs the same arithmetic in the same order must).   gpurun -- bash tests/tools/ab_bits.sh
set -u
for v in old new; do
  cp ab/libtwxhip_$v.so topowx_amd/libtwxhip.so
  python3 tests/tools/gpu_out_hash.py > gpurun_out/bits_$v.txt 2>&1
done
cp ab/libtwxhip_new.so topowx_amd/libtwxhip.so
if cmp -s gpurun_out/bits_old.txt gpurun_out/bits_new.txt; then echo "BITS EQUAL"; cat gpurun_out/bits_new.txt; else echo "BITS DIFFER"; diff gpurun_out/bits_old.txt gpurun_out/bits_new.txt; fi
