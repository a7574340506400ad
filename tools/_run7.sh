cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4g; mkdir -p $O; rm -f $O/*.txt
python -m pytest tests/test_gpu_conv.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -k "graph or raft or cfg4 or two_threads or default_path or determin or slow_fast or igev or evaluation" 2>&1 | tail -15 > $O/tests.txt
for i in 1 2 3; do
  python bench.py --skip-cpu-baseline --steps 20 --warmup 3 2>$O/bench_err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('max_abs_vs_reference'), j['roofline']['frac'])" >> $O/bench.txt
done
cat $O/tests.txt $O/bench.txt; tail -5 $O/bench_err.txt
