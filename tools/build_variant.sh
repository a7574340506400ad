#!/bin/bash
# Builds the library with extra -D flags on ONE source file (not conv2d.hip: see build_conv_variant.sh):
#   tools/build_variant.sh <name> <stem> "<flags>"   ->  dkt_stereo_amd/lib/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
[ -f dkt_stereo_amd/lib/libdktstereo.so ] || python -m dkt_stereo_amd.build >/dev/null
V=dkt_stereo_amd/lib/variants; O=dkt_stereo_amd/lib/obj
mkdir -p $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed"
name=$1; stem=$2; shift; shift
/opt/rocm/bin/hipcc $FL $@ -c dkt_stereo_amd/csrc/$stem.hip -o $V/${name}_$stem.o
others=$(ls $O/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $V/${name}_$stem.o -o $V/lib_$name.so
rm -f $V/${name}_$stem.o
