#!/bin/bash
# Build a variant of the library for a same-box A/B:  tests/tools/build_variant.sh NAME [-D...]  ->  ab/libtwxhip_NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p ab
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Itopowx_amd/csrc "$@" -o ab/libtwxhip_$name.so topowx_amd/csrc/twx_hip.hip
echo "built ab/libtwxhip_$name.so $*"
