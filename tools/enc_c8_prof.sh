#!/bin/bash
# rocprofv3 kernel trace of the encoder trunks (tools/enc_c8_check.py, ENC_ONLY): per-kernel time -> gpurun_out/enc_c8_prof.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/enc_prof; rm -rf $O; mkdir -p $O
ENC_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/tools/enc_c8_check.py > $O/log.txt 2>&1
python - <<PY > $R/gpurun_out/enc_c8_prof.txt
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print("%-100s %5s calls %10.1f us total %9.1f avg %9.1f min %9.1f max" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep -v "^W2026\|^E2026" $O/log.txt | tail -8 >> $R/gpurun_out/enc_c8_prof.txt
cat $R/gpurun_out/enc_c8_prof.txt
