#!/bin/bash
# Timing-only builds of the three-pass convolution unit (conv2d.hip + conv_ws.h) with extra -D flags:
#   tools/build_ws_variant.sh <name> "<flags>"  ->  dkt_stereo_amd/lib/variants/lib_<name>.so   (run with DKT_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
V=dkt_stereo_amd/lib/variants; O=dkt_stereo_amd/lib/obj
mkdir -p $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -Wno-pass-failed"
name=$1; shift
/opt/rocm/bin/hipcc $FL -DCONV_TU_PASSES=3 $@ -c dkt_stereo_amd/csrc/conv2d.hip -o $V/${name}_tu3.o
others=$(ls $O/*.o | grep -v "/conv2d_tu3.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $V/${name}_tu3.o -o $V/lib_$name.so
rm -f $V/${name}_tu3.o
