cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4d; mkdir -p $O; rm -f $O/*.txt
python -m pytest tests/test_gpu_conv.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -k "graph or raft or cfg4 or two_threads or default_path or determin or slow_fast or encoder or evaluation" 2>&1 | tail -15 > $O/tests.txt
for cfg in model.adopt_encoder_outputs=1 model.adopt_encoder_outputs=0 model.adopt_encoder_outputs=1 model.adopt_encoder_outputs=0; do
  echo "== $cfg" >> $O/bench.txt
  python tools/bench_cfg.py $cfg --skip-cpu-baseline --steps 20 --warmup 3 2>$O/bench_err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('max_abs_vs_reference'))" >> $O/bench.txt
done
echo "== B=8" >> $O/bench.txt
python bench.py --batch 8 --skip-cpu-baseline --steps 5 --warmup 2 2>>$O/bench_err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('max_abs_vs_reference'))" >> $O/bench.txt
cat $O/tests.txt $O/bench.txt; tail -5 $O/bench_err.txt
