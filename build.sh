#!/bin/bash
# Build libtwxhip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
# The compiler's per-kernel resource remarks (VGPRs / AGPRs / scratch / LDS / occupancy) are reduced into
# topowx_amd/libtwxhip.resources.txt: tests/test_isa_resources.py fails when a kriging / daily kernel spills or
# outgrows the register budget its waves_per_eu tuning assumes (a compiler bump would otherwise cost 10-20 % silently).
set -e
cd "$(dirname "$0")"
LOG=$(mktemp /tmp/twx_build.XXXXXX)
# stderr without the resource-usage remark blocks: a remark line and the source-context lines that follow IT are dropped;
# the context lines of genuine warnings / errors stay
show_diagnostics() {
    awk '/^In file included from/ { hold = hold $0 "\n"; next }
         /: remark: .*\[-Rpass-analysis=kernel-resource-usage\]/ { skip = 1; hold = ""; next }
         skip && (/^ *[0-9]+ \| / || /^ *\| /) { next }
         { skip = 0; printf "%s", hold; hold = ""; print }' "$1" >&2
}
if ! hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Iinclude -Itopowx_amd/csrc \
    -Rpass-analysis=kernel-resource-usage "$@" -o topowx_amd/libtwxhip.so topowx_amd/csrc/twx_hip.hip 2> "$LOG"; then
    show_diagnostics "$LOG"
    rm -f "$LOG"
    exit 1
fi
show_diagnostics "$LOG"
python3 tests/tools/isa_resources.py "$LOG" > topowx_amd/libtwxhip.resources.txt
rm -f "$LOG"
