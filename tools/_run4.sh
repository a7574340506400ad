cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4c; mkdir -p $O; rm -f $O/bench.txt
for cfg in model.encoder_order=0 model.encoder_order=1 model.encoder_order=2 model.encoder_order=0 model.encoder_order=1 model.encoder_order=2; do
  echo "== $cfg" >> $O/bench.txt
  python tools/bench_cfg.py $cfg --skip-cpu-baseline --steps 20 --warmup 3 2>$O/bench_err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" >> $O/bench.txt
done
cat $O/bench.txt
