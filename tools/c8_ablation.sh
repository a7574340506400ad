#!/bin/bash
# Timing-only ablation builds of conv_c8.hip: tools/c8_ablation.sh "1 2 8 ..."  -> dkt_stereo_amd/lib/variants/lib_c8abl<mask>.so
set -e
cd "$(dirname "$0")/.."
for m in $1; do tools/build_variant.sh c8abl$m conv_c8 -DC8_ABL=$m -mllvm -pragma-unroll-threshold=100000 & done
wait
