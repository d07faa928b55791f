#!/bin/bash
# Diagnostic: build libtwxhip with s_memtime stamps in k_uk<7>'s panel loop (-DTWX_UK_STAMP), run one bench step, reduce
# the stamps (tests/tools/uk_stamps.py), then rebuild the product library.  Run on the GPU box.
set -e
mkdir -p gpurun_out
./build.sh -DTWX_UK_STAMP
python3 bench.py --steps 1 --warmup 1 --no-daily --no-cpu-baseline --no-configs > /dev/null
python3 tests/tools/uk_stamps.py gpurun_out/uk_stamps.bin
./build.sh
