#!/bin/bash
# cfg4's per-GPU share (8 pairs per launch): scheduling and tile-shape variants.
# bash tools/b8_variants.sh [batch] -> gpurun_out/r05_b8_variants.txt  (pairs/s, ms per iteration, fused-GRU launch us)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
B=${1:-8}
O=gpurun_out/r05_b${B}_variants.txt
: > $O
run() {
  echo "## $1" >> $O
  env $1 timeout 600 python bench.py --steps 3 --warmup 1 --batch $B --skip-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pairs/s %.2f  ms/step %.1f  ms/iter %.3f  gru launch %.0f us  frac %.3f' % (d['value'], d['ms_per_step'], d['ms_per_iter'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" >> $O 2>&1
}
run "DKT_NOP=1"
run "DKT_C8_CFG=zr16=1,q16=2"
run "DKT_C8_CFG=zr16=2,q16=2"
run "DKT_C8_CFG=zr16=3,q16=3"
run "DKT_C8_CFG=c2=3"
run "DKT_C8_CFG=enc=2"
run "DKT_C8_CFG=zr16=1,q16=2,c2=3"
run "DKT_NOP=2"
run "DKT_C8_FORK=0"
run "DKT_ENCODER_STREAMS=0"
cat $O
