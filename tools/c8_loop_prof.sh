#!/bin/bash
# rocprofv3 kernel trace of the C8S loop at 736x1248 / 32 iterations: per-kernel time per pair -> gpurun_out/c8_loop_prof.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c8_loop_prof; rm -rf $O; mkdir -p $O
C8_ONLY=${C8_ONLY:-1} timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/c8_loop_check.py > $O/log.txt 2>&1
python - <<PY > $R/gpurun_out/c8_loop_prof.txt
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last forward = after the last convex_upsample but one
ups = [i for i, r in enumerate(rows) if "convex_upsample" in r["Kernel_Name"]]
lo, hi = ups[-2] + 1, ups[-1] + 1
sel = rows[lo:hi]
acc = collections.OrderedDict()
for r in sel:
    k = r["Kernel_Name"][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print("last pair: %d dispatches, wall %.1f us, kernel time %.1f us" % (len(sel), wall, sum(a[1] for a in acc.values())))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %4d %10.1f us  %8.1f avg" % (k, n, t, t / n))
PY
cat $R/gpurun_out/c8_loop_prof.txt
