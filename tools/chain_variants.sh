#!/bin/bash
# The chain launches (DKT_C8_CHAIN: 0 off, 1 on, 2 timing only = no waits, the fusion's upper bound) at batch 1 and 8.
# bash tools/chain_variants.sh -> gpurun_out/r05_chain.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05_chain.txt
: > $O
run() {
  echo "## $1   batch $2" >> $O
  env $1 timeout 600 python bench.py --steps $3 --warmup 2 --batch $2 --skip-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pairs/s %.2f  ms/step %.2f  ms/iter %.4f  gru launch %.0f us  max|d| vs fixture %s' % (d['value'], d['ms_per_step'], d['ms_per_iter'], d['roofline']['avg_launch_us'], d.get('max_abs_vs_reference')))" >> $O 2>&1
}
for rep in 1 2; do
run "DKT_C8_CHAIN=0" 1 20
run "DKT_C8_CHAIN=1" 1 20
run "DKT_C8_CHAIN=2" 1 20
run "DKT_C8_CHAIN=1 DKT_C8_CHAIN_BLOCKS=512" 1 20
done
run "DKT_C8_CHAIN=0" 8 4
run "DKT_C8_CHAIN=1" 8 4
run "DKT_C8_CHAIN=2" 8 4
cat $O
