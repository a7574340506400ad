#!/bin/bash
# Counters of the conv_c8 launch that is almost only epilogue (16 input channels, gate epilogue): bash tools/c8_epi_pmc.sh -> gpurun_out/c8_epi_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c8_epi_pmc; rm -rf $O; mkdir -p $O
cat > /tmp/epi_probe.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from dkt_stereo_amd import conv_c8 as c8
torch.manual_seed(0)
H, W = 184, 312
with torch.no_grad():
    h = torch.tanh(torch.randn(1, 128, H, W, device="cuda:0"))
    cz, cr = (torch.randn(1, 128, H, W, device="cuda:0") for _ in range(2))
    rh = c8.ActC8(1, 128, H, W, "cuda:0")
    a = c8.pack(torch.randn(1, 16, H, W, device="cuda:0"))
    zr = torch.nn.Conv2d(16, 256, 3, padding=1).cuda()
    for _ in range(6):
        c8.gate_zr([a], zr, cz, cr, h, rh_c8=rh, cfg=1)
    torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TD_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python /tmp/epi_probe.py > $O/p$i.log 2>&1
done
python - <<PY > $R/gpurun_out/c8_epi_pmc.txt
import csv, glob, collections
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_c8_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("   %-32s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $R/gpurun_out/c8_epi_pmc.txt
