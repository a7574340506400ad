#!/bin/bash
# Builds the library with extra -D flags on the convolution kernel only:
#   tools/build_conv_variant.sh <name> "<flags>"   ->  dkt_stereo_amd/lib/variants/lib_<name>.so
# (select it with DKT_LIB_PATH; tools/conv_ablation.py times every library in that directory).
set -e
cd "$(dirname "$0")/.."
[ -f dkt_stereo_amd/lib/libdktstereo.so ] || python -m dkt_stereo_amd.build >/dev/null
V=dkt_stereo_amd/lib/variants; O=dkt_stereo_amd/lib/obj
mkdir -p $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed"
name=$1; shift
# the 3-pass kernels (the default backend) live in translation unit 3
/opt/rocm/bin/hipcc $FL $@ -DCONV_TU_PASSES=3 -c dkt_stereo_amd/csrc/conv2d.hip -o $V/${name}_tu3.o
others=$(ls $O/*.o | grep -v conv2d_tu3)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $V/${name}_tu3.o -o $V/lib_$name.so
rm -f $V/${name}_tu3.o
ls -la $V/lib_$name.so
