#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only) of the loop's own
# launches (tools/pmc/pmc_probe.py), with known-traffic calibration kernels in the same passes.
# Run on the GPU box:  bash tools/pmc/run_pmc.sh [batch]  -> gpurun_out/r06_pmc[_b<batch>]/{summary_tail.txt,traffic.json}
# (bench.py --pmc calls it)
set -e
B=${1:-1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/pmc/pmc_calib.hip -o /tmp/libpmc_calib.so
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06_pmc; [ "$B" != 1 ] && O=${O}_b$B
rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  PYTHONPATH=$R timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o $c -- python $R/tools/pmc/pmc_probe.py $B > $O/$c.log 2>&1
done
cd $R
python tools/pmc/pmc_summary.py $O | sed 's/round 2/round 6/' > $O/summary.txt
grep -E "^#|calib_|gru_c8_kernel|motion_front_kernel|corr1d_lookup_skew_kernel" $O/summary.txt | awk '/gru_c8_kernel/{g++; if (g>40) next} /motion_front_kernel/{m++; if (m>40) next} {print}' > $O/summary_tail.txt
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE        # (the raw counter CSVs are tens of MB; summary + pmc.json stay)
python tools/pmc/make_traffic.py $O $B
