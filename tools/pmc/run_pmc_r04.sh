#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only) of the round-4
# kernels bench.py reports, with known-traffic calibration kernels in the same passes.
# Run on the GPU box:  bash tools/pmc/run_pmc_r04.sh   -> gpurun_out/r04_pmc/{summary.txt,traffic.json}   (bench.py --pmc calls it)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $R/tools/pmc/pmc_calib.hip -o /tmp/libpmc_calib.so
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04_pmc
rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  PYTHONPATH=$R timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o $c -- python $R/tools/pmc/pmc_probe_r04.py > $O/$c.log 2>&1
done
cd $R
python tools/pmc/pmc_summary.py $O | sed 's/round 2/round 4/' > $O/summary.txt
python tools/pmc/make_traffic_r04.py $O
