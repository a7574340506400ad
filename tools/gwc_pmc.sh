#!/bin/bash
# MFMA-pipe counters of the group-wise correlation kernel (gwc_mfma.hip): bash tools/gwc_pmc.sh -> gpurun_out/gwc_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/gwc_pmc; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p1 -- python $R/tools/gwc_check.py > $O/p1.log 2>&1
python - <<PY > $R/gpurun_out/gwc_pmc.txt
import csv, glob, collections
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/gwc_check.py")
print("# per dispatch averages; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) as the guide defines it")
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gwc" not in k: continue
        acc[(k[:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), d in acc.items():
        print(k, "grid", g)
        for c, v in sorted(d.items()):
            print("   %-28s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
        m, b = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("SQ_BUSY_CU_CYCLES")
        if m and b:
            print("   MFMA busy / (4 x CU busy)     %15.1f %%" % (100.0 * (sum(m) / len(m)) / (4 * sum(b) / len(b))))
PY
cat $R/gpurun_out/gwc_pmc.txt
