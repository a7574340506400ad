#!/bin/bash
# Round 3 evidence on a GPU box -> gpurun_out/ (copy what is to be kept into profiles/):  bash tools/collect_profiles_r04.sh [steps...]
# steps: tests prof bench micro trace stress stages pmc ws gwc   (default: all but gwc)
R=r04
O=gpurun_out
STEPS=${@:-tests prof bench micro trace stress stages pmc ws}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
for s in $STEPS; do case $s in
tests) python -m pytest tests -q -m gpu 2>&1 | grep -v -E "^(RCCL|HIP|ROCm) version|^Hostname|^Librccl|amdgpu.ids" | tail -5 > $O/${R}_gputests.txt ;;
prof)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline > $O/${R}_prof_bench.log 2>&1
  T=$(find $O/${R}_prof -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_pair_breakdown.py $T --pair 6 --phases --timeline 10 --encoders > $O/${R}_pair_breakdown.txt 2>&1
  python tools/rocprof_summary.py $T > $O/${R}_kernels.txt 2>&1
  cp $(find $O/${R}_prof -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
  rm -rf $O/${R}_prof ;;
bench)
  python bench.py --steps 20 --warmup 3 --pmc 2>/dev/null | tail -1 > $O/${R}_bench_b1.json      # (--pmc: roofline.traffic measured in this run)
  DKT_FUSE_LOOKUP=0 python bench.py --steps 3 --warmup 1 --conv-backend miopen --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_fp32.json      # true-fp32 (vendor) convolutions
  python bench.py --steps 5 --warmup 2 --batch 8 --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_b8.json
  python bench.py --steps 10 --warmup 3 --conv-backend f16 --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_f16.json
  python tools/bench_configs.py 2>/dev/null | grep "^{" > $O/${R}_bench_configs.jsonl ;;
micro) python tools/bench_kernels.py c8 lookup build volumes next 2>&1 | grep -v amdgpu > $O/${R}_kernel_microbench.txt ;;
trace) python tools/gru_c8_trace.py --rebuild 2>&1 | grep -v amdgpu > $O/${R}_gru_c8_phases.txt ;;
stress) python tools/stress_forward.py 200 2>&1 | grep -v amdgpu > $O/${R}_stress_forward.txt ;;
stages) python tools/iteration_stages.py 2>&1 | grep -v amdgpu > $O/${R}_iteration_stages.txt ;;
pmc) bash tools/pmc/run_pmc_r04.sh > $O/${R}_pmc.log 2>&1; python tools/pmc/make_traffic_r04.py $O/r04_pmc --profiles >> $O/${R}_pmc.log 2>&1 ;;
ws) (python tools/conv_ws_check.py --time; DKT_CONV_WS=0 python tools/conv_ws_check.py --time) 2>&1 | grep -v amdgpu > $O/${R}_conv_ws.txt ;;
gwc) bash tools/gwc_pmc.sh > /dev/null 2>&1 ;;
esac; done
ls -la $O | tail -20
