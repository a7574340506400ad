#!/bin/bash
# Collects the round's evidence on a GPU box into gpurun_out/ (copy what is to be kept into profiles/):
#   tools/collect_profiles.sh r02
# GPU test suite, rocprofv3 kernel trace of the default bench command + per-pair breakdown, bench lines (B=1 with the CPU
# baseline, B=8), the other BASELINE.json configs, kernel micro-benchmarks, stand-alone stage times.
R=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/${R}_gputests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline > $O/${R}_prof_bench.log 2>&1
T=$(find $O/${R}_prof -name "*kernel_trace.csv" | head -1)
python tools/rocprof_pair_breakdown.py $T --pair 3 --phases --timeline 10 --encoders > $O/${R}_pair_breakdown.txt 2>&1
python tools/rocprof_summary.py $T > $O/${R}_kernels.txt 2>&1
rm -rf $O/${R}_prof
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${R}_bench_b1.json
python bench.py --steps 5 --warmup 2 --batch 8 --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_b8.json
python tools/bench_configs.py 2>/dev/null | grep "^{" > $O/${R}_bench_configs.jsonl
python tools/bench_kernels.py lookup build volumes next 2>&1 | grep -v amdgpu > $O/${R}_kernel_microbench.txt
python tools/iteration_stages.py 2>&1 | grep -v amdgpu > $O/${R}_iteration_stages.txt
