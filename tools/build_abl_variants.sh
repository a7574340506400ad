#!/bin/bash
# Timing-only ablation builds of the convolution kernel (conv2d.hip CONV_ABL bit mask: 1 = no weight loads,
# 2 = no LDS fragment reads, 4 = no staging, 8 = no MFMAs) linked against the product's other objects.
# Output: dkt_stereo_amd/lib/variants/lib_abl<mask>.so (results of these builds are WRONG by construction).
set -e
cd "$(dirname "$0")/.."
python -m dkt_stereo_amd.build >/dev/null
V=dkt_stereo_amd/lib/variants; O=dkt_stereo_amd/lib/obj
mkdir -p $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $FL -DCONV_ABL=$m -DCONV_TU_PASSES=3 -c dkt_stereo_amd/csrc/conv2d.hip -o $V/abl${m}_tu3.o
    others=$(ls $O/*.o | grep -v conv2d_tu3)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $V/abl${m}_tu3.o -o $V/lib_abl$m.so
    rm -f $V/abl${m}_tu3.o ) &
done
wait
ls -la $V
