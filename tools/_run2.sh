cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4b; mkdir -p $O
python -m pytest tests/test_gpu_round4.py -q -x -k "motion_front or resample" 2>&1 | tail -15 > $O/tests.txt
python -m pytest tests/test_gpu_round3.py -q -x -k "raft_c8_loop or igev_c8_loop or producers" 2>&1 | tail -15 >> $O/tests.txt
python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -k "raft_stereo_end_to_end or cfg4 or two_threads" 2>&1 | tail -8 >> $O/tests.txt
for cfg in FRONT=1 FRONT=0 FRONT=1; do
  echo "== $cfg" >> $O/bench.txt
  python tools/bench_cfg.py $cfg --skip-cpu-baseline --steps 20 --warmup 3 2>$O/bench_err_$cfg.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('roofline',{}).get('frac'), j.get('max_abs_vs_reference'), j.get('accuracy'))" >> $O/bench.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline > $O/prof_bench.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/rocprof_pair_breakdown.py $T --pair 3 --phases --timeline 10 --encoders > $O/pair_breakdown.txt 2>&1
rm -rf $O/prof
cat $O/tests.txt $O/bench.txt; head -16 $O/pair_breakdown.txt
