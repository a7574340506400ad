#!/bin/bash
# Tile shapes of the window's layers, same box, alternating:  bash tools/cfg_variants.sh [batch]   (DKT_C8_CFG, loop_c8._env_cfg)
B=${1:-1}
for r in 1 2; do
for v in "" "c2=3" "zr16=3" "c2=3,zr16=3" "q16=3"; do
  DKT_C8_CFG=$v timeout 600 python bench.py --steps 20 --warmup 3 --batch $B --skip-cpu-baseline --distinct-pairs 0 2>/dev/null | tail -1 | \
    python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg %-14s %.2f pairs/s  %.3f ms/iter' % ('$v' or 'default', j['value'], j['ms_per_iter']))"
done; done
