cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/s4b; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline > $O/prof_bench.log 2>&1
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/rocprof_pair_breakdown.py $T --pair 3 --phases --timeline 10 --encoders > $O/pair_breakdown.txt 2>&1
rm -rf $O/prof
head -32 $O/pair_breakdown.txt
