#!/usr/bin/env python3
"""profiles/r03_hbm_traffic.{txt,json} from gpurun_out/r03_pmc (tools/pmc/run_pmc_r03.sh) and the in-pipeline lookup time of
profiles/r03_pair_breakdown.txt:  python tools/pmc/make_traffic_r03.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pmc = json.load(open(os.path.join(ROOT, "gpurun_out", "r03_pmc", "pmc.json")))["rows"]
KB = 1024


def avg(key, which, lo=0, hi=None):
    v = pmc[key][which][lo:hi]
    return sum(v) / len(v)


zr = "conv_c8_kernel<4, 2, 4, 4>"
q = "conv_c8_kernel<2, 4, 2, 4>"
lk = "corr_feat64_kernel<4>"
zr_f, zr_w = 2 * KB * avg(zr, "fetch_kb"), KB * avg(zr, "write_kb")
q_f, q_w = 2 * KB * avg(q, "fetch_kb"), KB * avg(q, "write_kb")
# corr_feat64 dispatch order: B=1 C8S x3, B=1 NCHW x3, B=8 C8S x3, B=8 NCHW x3
l1_f, l1_w = 2 * KB * avg(lk, "fetch_kb", 0, 3), KB * avg(lk, "write_kb", 0, 3)
l1n_f, l1n_w = 2 * KB * avg(lk, "fetch_kb", 3, 6), KB * avg(lk, "write_kb", 3, 6)
l8_f, l8_w = 2 * KB * avg(lk, "fetch_kb", 6, 9), KB * avg(lk, "write_kb", 6, 9)
l8n_f, l8n_w = 2 * KB * avg(lk, "fetch_kb", 9, 12), KB * avg(lk, "write_kb", 9, 12)
# in-pipeline lookup: corr_feat64_kernel row of the GRU-loop table in the pair breakdown
inpipe, rng = None, ""
for line in open(os.path.join(ROOT, "profiles", "r03_pair_breakdown.txt")):
    m = re.match(r"corr_feat64_kernel<4>\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
    if m and int(m.group(1)) == 32:
        inpipe, rng = float(m.group(3)), "%s-%s us" % (m.group(4), m.group(5))
        break
comp = 238.7e6
hdr = """# HBM-side traffic of the kernels bench.py reports (MI355X, round 3): bash tools/pmc/run_pmc_r03.sh (tools/pmc/pmc_probe_r03.py)
#   rocprofv3 --pmc FETCH_SIZE  --kernel-trace -- python tools/pmc/pmc_probe_r03.py      (pass 1)
#   rocprofv3 --pmc WRITE_SIZE  --kernel-trace -- python tools/pmc/pmc_probe_r03.py      (pass 2)
# Counter values as reported (KB = 1024 B), per dispatch, 3 dispatches per shape.  corr_feat64_kernel rows in dispatch order:
# B=1 C8S output x3, B=1 fp32 NCHW output x3, B=8 C8S x3, B=8 NCHW x3.  conv_c8_kernel<4,2,4,4> = gru08 z|r 384->256 + gate
# epilogue (tile shape 1); <2,4,2,4> = gru08 q 384->128 + state-update epilogue (tile shape 2); both 184x312, C8S operands.
# CALIBRATION in the same passes (tools/pmc/pmc_calib.hip, 1 GiB streams = 4x the Infinity Cache): FETCH_SIZE reports exactly
# 0.500 of the bytes read for 4-B/lane AND 16-B/lane coalesced streams, WRITE_SIZE exactly 1.000 of the bytes written:
# every read figure quoted in DESIGN.md / bench.py is 2 x FETCH_SIZE.  Infinity-Cache hits are counted (memory-side counters).
#
# Summary (bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE):
#   gru08 z|r conv_c8   : %.1f MB read + %.1f MB written = %.1f MB  vs 238.7 MB compulsory (88.2 C8S operands + 88.2 gate
#                         operands cz, cr, h + 3.5 weights + 58.8 outputs z, r*h)  = %.2fx   (round 2: 554.1 MB = 2.32x)
#   gru08 q conv_c8     : %.1f MB read + %.1f MB written = %.1f MB  (88.2 operands + 88.2 cq, z, h + 1.8 weights + 58.8 h' as fp32 and C8S = 237.0 MB compulsory: %.2fx)
#   lookup+convc1 -> C8S: B=1 %.1f + %.1f = %.1f MB vs 24.1 MB algorithmic = %.2fx;  B=8 %.1f + %.1f = %.1f MB vs 192.9 = %.2fx
#   in the pipeline (rocprofv3 --kernel-trace of bench.py, profiles/r03_pair_breakdown.txt): corr_feat64_kernel %s us average
#   (%s) over the 32 dispatches of a timed pair.
""" % (zr_f / 1e6, zr_w / 1e6, (zr_f + zr_w) / 1e6, (zr_f + zr_w) / comp, q_f / 1e6, q_w / 1e6, (q_f + q_w) / 1e6, (q_f + q_w) / 237.0e6,
       l1_f / 1e6, l1_w / 1e6, (l1_f + l1_w) / 1e6, (l1_f + l1_w) / 24111360, l8_f / 1e6, l8_w / 1e6, (l8_f + l8_w) / 1e6,
       (l8_f + l8_w) / 192890880, inpipe, rng)
src = open(os.path.join(ROOT, "gpurun_out", "r03_pmc", "summary.txt")).read().splitlines()
keep = [ln for ln in src if re.match(r"^(#|kernel|calib_|conv_c8_kernel|corr_feat64|corr1d_skew|corr1d_build)", ln)]
open(os.path.join(ROOT, "profiles", "r03_hbm_traffic.txt"), "w").write(hdr + "\n".join(keep) + "\n")
j = {
    "source": "profiles/r03_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch; FETCH_SIZE x2: calibrated 0.500 of the bytes read on 1 GiB known-traffic streams at 4 and 16 B/lane in the same passes; WRITE_SIZE calibrated 1.000)",
    "fetch_correction": 2.0,
    "conv_zr_gate_bytes": int(zr_f + zr_w), "conv_zr_gate_fetch_bytes": int(zr_f), "conv_zr_gate_write_bytes": int(zr_w),
    "conv_zr_gate_compulsory_bytes": int(comp), "conv_q_gate_bytes": int(q_f + q_w),
    "lookup_conv1x1_b1_bytes": int(l1_f + l1_w), "lookup_conv1x1_b8_bytes": int(l8_f + l8_w),
    "lookup_conv1x1_nchw_b1_bytes": int(l1n_f + l1n_w), "lookup_conv1x1_nchw_b8_bytes": int(l8n_f + l8n_w),
    "lookup_b1_bytes": 21039786, "lookup_b8_bytes": 168600842,
    "lookup_in_pipeline_us": inpipe,
    "lookup_in_pipeline_source": "profiles/r03_pair_breakdown.txt: corr_feat64_kernel<4>, 32 dispatches of one timed 736x1248 pair (%s)" % rng,
}
json.dump(j, open(os.path.join(ROOT, "profiles", "r03_hbm_traffic.json"), "w"), indent=1)
print(hdr)
