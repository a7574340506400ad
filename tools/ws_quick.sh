#!/bin/bash
# quick compile of the ws kernel alone + resource usage:  tools/ws_quick.sh [extra flags]
cd /tmp/wsq && time /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" -c ws_only.hip -o ws_only.o 2>&1 | grep -E "Function Name|VGPRs|AGPRs|Scratch|Spill|error" | grep -v "SGPRs:"
