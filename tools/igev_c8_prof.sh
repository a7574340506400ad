#!/bin/bash
# rocprofv3 kernel trace of the IGEV loop (cfg3, tools/bench_configs.py first line) -> gpurun_out/igev_c8_prof.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/igev_prof; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/tools/bench_configs.py cfg3 > $O/log.txt 2>&1
python - <<PY > $R/gpurun_out/igev_c8_prof.txt
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print("%-90s %5s calls %10.1f us total %9.1f avg" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
grep "^{" $O/log.txt | head -1 | cut -c1-300 >> $R/gpurun_out/igev_c8_prof.txt
cat $R/gpurun_out/igev_c8_prof.txt
