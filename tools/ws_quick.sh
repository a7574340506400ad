#!/bin/bash
# Seconds-long compile of conv_ws.h's kernels alone (no other instantiation of conv2d.hip) with the register / spill report:
#   tools/ws_quick.sh [extra hipcc flags]        -> /tmp/wsq/ws_only.o   (tools/ws_isa.sh /tmp/wsq/ws_only.o for the instruction mix)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/wsq
cat > /tmp/wsq/ws_only.hip <<EOT
#define CONV_TU_PASSES 99
#include "$R/dkt_stereo_amd/csrc/conv2d.hip"
#include "$R/dkt_stereo_amd/csrc/conv_ws.h"
int ws_only_entry(ConvArgs a, int B, hipStream_t st) { return launch_conv_ws(a, B, st); }
EOT
cd /tmp/wsq && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -Wno-pass-failed -Rpass-analysis=kernel-resource-usage "$@" -c ws_only.hip -o ws_only.o 2>&1 | grep -E "Function Name|VGPRs|AGPRs|Scratch|Spill|error" | grep -v "SGPRs:"
