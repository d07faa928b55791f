#!/bin/bash
# Build libtwxhip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Itopowx_amd/csrc \
    "$@" -o topowx_amd/libtwxhip.so topowx_amd/csrc/twx_hip.hip
