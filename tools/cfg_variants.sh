#!/bin/bash
# Tile-shape variants of the loop's layer classes (DKT_C8_CFG) at a batch size: bash tools/cfg_variants.sh <batch> <steps> "<cfg1>" "<cfg2>" ...
# -> gpurun_out/r05_cfg_b<batch>.txt  (every variant twice, alternating with the default)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
B=$1; S=$2; shift 2
O=gpurun_out/r05_cfg_b${B}.txt
: > $O
run() {
  echo "## DKT_C8_CFG=$1" >> $O
  DKT_C8_CFG="$1" timeout 600 python bench.py --steps $S --warmup 2 --batch $B --skip-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pairs/s %.2f  ms/step %.2f  ms/iter %.4f  gru launch %.0f us  max|d| %s' % (d['value'], d['ms_per_step'], d['ms_per_iter'], d['roofline']['avg_launch_us'], d.get('max_abs_vs_reference')))" >> $O 2>&1
}
for rep in 1 2; do
  run ""
  for c in "$@"; do run "$c"; done
done
cat $O
