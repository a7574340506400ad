cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_enc/$n -o $n -- python $R/tools/prof_conv.py enc 3 > $R/gpurun_out/pmc_enc/$n.log 2>&1; }
mkdir -p $R/gpurun_out/pmc_enc
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run p2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run p3 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
run p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $R; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_enc/*/*counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv2d_f16s" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("%-32s %16.0f  (n=%d)"%(k,v[-1],len(v)))
PY
