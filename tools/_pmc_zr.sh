cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_${PMC_TARGET:-zr_gate}
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${PMC_TARGET:-zr_gate}/$n -o $n -- python $R/tools/prof_conv.py ${PMC_TARGET:-zr_gate} 3 > $R/gpurun_out/pmc_${PMC_TARGET:-zr_gate}/$n.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
run p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
cd $R; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_${PMC_TARGET:-zr_gate}/*/*counter_collection.csv")):
    d=collections.defaultdict(list); t={}
    for r in csv.DictReader(open(f)):
        if "conv2d_f16s" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"])); t[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for k,v in d.items(): print("%-30s %16.0f"%(k,v[-1]))
    print("  kernel us:", list(t.values()))
PY
