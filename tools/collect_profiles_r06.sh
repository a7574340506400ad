#!/bin/bash
# Round 6 evidence on a GPU box -> gpurun_out/ (copy what is to be kept into profiles/):  bash tools/collect_profiles_r06.sh [steps...]
# steps: tests prof bench b8 trace stress pmc sched schedline configs smoke mixed gaps cumask window   (default: tests prof bench b8 trace)
R=r06
O=gpurun_out
STEPS=${@:-tests prof bench b8 trace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$GRAFT_REPO_ROOT
F='^(RCCL|HIP|ROCm) version|^Hostname|^Librccl|amdgpu.ids'
for s in $STEPS; do case $s in
tests) timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v -E "$F" | tail -15 > $O/${R}_gputests.txt ;;
prof)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof -o bench -- python bench.py --steps 3 --warmup 2 --skip-cpu-baseline --distinct-pairs 0 > $O/${R}_prof_bench.log 2>&1
  T=$(find $O/${R}_prof -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_pair_breakdown.py $T --pair 6 --phases --timeline 10 --encoders > $O/${R}_pair_breakdown.txt 2>&1
  python tools/rocprof_summary.py $T > $O/${R}_kernels.txt 2>&1
  cp $(find $O/${R}_prof -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
  rm -rf $O/${R}_prof ;;
bench) timeout 900 python bench.py --steps 20 --warmup 3 2>$O/${R}_bench_b1.err | tail -1 > $O/${R}_bench_b1.json ;;
b8) timeout 900 python bench.py --steps 5 --warmup 2 --batch 8 --skip-cpu-baseline 2>$O/${R}_bench_b8.err | tail -1 > $O/${R}_bench_b8.json ;;
trace) timeout 600 python tools/gru_c8_trace.py --batch=1 --batch=8 2>&1 | grep -v amdgpu > $O/${R}_gru_c8_phases.txt
  for p in 2 1; do timeout 600 python tools/gru_c8_trace.py --batch=1 --passes=$p 2>&1 | grep -v amdgpu >> $O/${R}_gru_c8_phases.txt; done ;;
mixed) timeout 600 python bench.py --steps 20 --warmup 3 --mixed-precision --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_mixed.json ;;
gaps) for m in default unchecked; do rocprofv3 --kernel-trace --output-format csv -d $O/${R}_gap_$m -o t -- python tools/default_mode_probe.py $m > $O/${R}_gap_$m.log 2>&1
    T=$(find $O/${R}_gap_$m -name "*kernel_trace.csv" | head -1); (grep "ms per pair" $O/${R}_gap_$m.log; python tools/idle_gaps.py $T --pair -3 --min 8) > $O/${R}_idle_gaps_$m.txt; rm -rf $O/${R}_gap_$m; done ;;
cumask) timeout 600 python tools/cumask_ab.py 0 224 192 -28 -24 2>&1 | grep -v amdgpu > $O/${R}_cumask.txt ;;
window) timeout 600 python tools/window_kernels.py 1 8 2>&1 | grep -v amdgpu > $O/${R}_window_kernels.txt
  # the same kernels without their epilogues (tools/c8_variant.py --tag=noepi, built in the container): the most ANY overlap of an
  # item's epilogue with the next item's MFMAs could hide
  if [ -f tools/_build/libdktstereo_c8noepi.so ]; then (echo; echo "# ---- timing only: every tile's epilogue skipped (tools/c8_variant.py --tag=noepi)"; DKT_LIB_PATH=$PWD/tools/_build/libdktstereo_c8noepi.so timeout 600 python tools/window_kernels.py 1 2>&1 | grep -v amdgpu) >> $O/${R}_window_kernels.txt; fi ;;
stress) timeout 1500 python tools/stress_forward.py 1000 2>&1 | grep -v amdgpu > $O/${R}_stress_forward.txt ;;
pmc) bash tools/pmc/run_pmc.sh 1 > $O/${R}_pmc.log 2>&1; python tools/pmc/make_traffic.py $O/r06_pmc 1 --profiles >> $O/${R}_pmc.log 2>&1 ;;
sched) timeout 2400 python tools/precision_schedule.py 2>&1 | grep -v amdgpu > $O/${R}_precision_schedule.txt ;;
schedline) timeout 600 python bench.py --steps 20 --warmup 3 --schedule 0,16 --skip-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_sched.json ;;
configs) timeout 900 python tools/bench_configs.py 2>/dev/null | grep "^{" > $O/${R}_bench_configs.jsonl ;;
smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3 > $O/${R}_smoke.txt ;;
esac; done
ls -la $O | tail -20
