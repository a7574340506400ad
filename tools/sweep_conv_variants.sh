#!/bin/bash
# Builds scheduling variants of conv2d.hip into gpurun_out-independent lib/variants/ (local), then
# `python tools/_exp_variants.py` (on the GPU box) times the flagship layers with each.
set -e
cd "$(dirname "$0")/.."
mkdir -p dkt_stereo_amd/lib/variants
build() { # name, flags...
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-pass-failed "$@" \
     dkt_stereo_amd/csrc/*.hip -o dkt_stereo_amd/lib/variants/lib_$name.so &
}
build base
build ar2 -DCONV_AR=2
build valu2 -DCONV_SGB_VALU=2
build valu5 -DCONV_SGB_VALU=5
wait
build mem2 -DCONV_SGB_MEM=2
build mfma2 -DCONV_SGB_MFMA=2 -DCONV_SGB_MEM=2 -DCONV_SGB_VALU=6
build mfma4 -DCONV_SGB_MFMA=4 -DCONV_SGB_MEM=4 -DCONV_SGB_VALU=12
build nosgb -DCONV_SGB_MFMA=1 -DCONV_SGB_MEM=0 -DCONV_SGB_VALU=0
wait
ls -la dkt_stereo_amd/lib/variants
