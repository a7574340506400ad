#!/bin/bash
# A/B of environment switches with bench.py: bash tools/env_variants.sh <batch> <steps> <out-name> "<VAR=val ...>" ... (default first, twice)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
B=$1; S=$2; N=$3; shift 3
O=gpurun_out/$N
: > $O
run() {
  echo "## $1   (batch $B)" >> $O
  env $1 timeout 600 python bench.py --steps $S --warmup 2 --batch $B --skip-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pairs/s %.2f  ms/step %.2f  ms/iter %.4f  gru launch %.0f us  max|d| %s' % (d['value'], d['ms_per_step'], d['ms_per_iter'], d['roofline']['avg_launch_us'], d.get('max_abs_vs_reference')))" >> $O 2>&1
}
for rep in 1 2; do
  run "DKT_NOP=1"
  for c in "$@"; do run "$c"; done
done
cat $O
