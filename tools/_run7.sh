cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4g; mkdir -p $O; rm -f $O/*.txt
for cfg in PROLOGUE_FORK=1 PROLOGUE_FORK=0 PROLOGUE_FORK=1 PROLOGUE_FORK=0 PROLOGUE_FORK=1 PROLOGUE_FORK=0; do
  echo "== $cfg" >> $O/bench.txt
  python tools/bench_cfg.py $cfg --skip-cpu-baseline --steps 20 --warmup 3 2>$O/bench_err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('max_abs_vs_reference'), j['roofline']['frac'])" >> $O/bench.txt
done
cat $O/bench.txt; tail -5 $O/bench_err.txt
