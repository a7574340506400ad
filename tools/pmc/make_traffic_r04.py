#!/usr/bin/env python3
"""traffic.json (and, with --profiles, profiles/r04_hbm_traffic.{txt,json}) from a tools/pmc/run_pmc_r04.sh output directory:
bytes per launch = FETCH_SIZE / (calibrated fraction of the bytes read) + WRITE_SIZE / (calibrated fraction written)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
pmc = json.load(open(os.path.join(out, "pmc.json")))
rows, fac = pmc["rows"], pmc["factors"]
KB = 1024
ff = fac.get("calib_read16.fetch", 0.5)          # FETCH_SIZE reports this fraction of the bytes read (0.500 on gfx950)
wf = fac.get("calib_write4.write", 1.0)


def avg(key, which, lo=0, hi=None):
    v = rows[key][which][lo:hi]
    return sum(v) / len(v)


gk = [k for k in rows if k.startswith("gru_c8_kernel")][0]
lk = [k for k in rows if k.startswith("corr_feat64_kernel")][0]
g_f, g_w = KB * avg(gk, "fetch_kb") / ff, KB * avg(gk, "write_kb") / wf
l1_f, l1_w = KB * avg(lk, "fetch_kb", 0, 3) / ff, KB * avg(lk, "write_kb", 0, 3) / wf
l8_f, l8_w = KB * avg(lk, "fetch_kb", 3, 6) / ff, KB * avg(lk, "write_kb", 3, 6) / wf
# compulsory bytes of one fused ConvGRU step at 184x312 (+ 23x39): C8S operands h, x1, x2 read by z|r (88.2 MB) and x1, x2, r*h
# by q (88.2), context terms cz, cr, cq (88.2), state h read once (29.4), weights 3.5 + 1.8, written r*h (29.4) and h' as fp32
# and C8S (58.8); the coarse level adds 0.6 %
comp = (88.2 + 88.2 + 88.2 + 29.4 + 5.3 + 29.4 + 58.8) * 1e6 * 1.006
j = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch (tools/pmc/run_pmc_r04.sh); reads = FETCH_SIZE / %.3f, "
              "writes = WRITE_SIZE / %.3f from 1 GiB known-traffic streams in the same passes" % (ff, wf),
    "fetch_fraction": ff, "write_fraction": wf,
    "gru_bytes": int(g_f + g_w), "gru_fetch_bytes": int(g_f), "gru_write_bytes": int(g_w), "gru_compulsory_bytes": int(comp),
    "lookup_conv1x1_b1_bytes": int(l1_f + l1_w), "lookup_conv1x1_b8_bytes": int(l8_f + l8_w),
}
json.dump(j, open(os.path.join(out, "traffic.json"), "w"), indent=1)
txt = """# HBM-side traffic of the kernels bench.py reports (MI355X, round 4): bash tools/pmc/run_pmc_r04.sh
#   fused ConvGRU launch (gru_c8_kernel, gru08 184x312 + gru32 23x39): %.1f MB read + %.1f MB written = %.1f MB per launch
#       vs %.1f MB compulsory = %.2fx   (round 3, the two launches it replaces: 389.1 + 393.7 = 782.8 MB)
#   lookup+convc1 -> C8S: B=1 %.1f MB vs 24.1 MB algorithmic = %.2fx;  B=8 %.1f MB vs 192.9 = %.2fx
""" % (g_f / 1e6, g_w / 1e6, (g_f + g_w) / 1e6, comp / 1e6, (g_f + g_w) / comp,
       (l1_f + l1_w) / 1e6, (l1_f + l1_w) / 24111360, (l8_f + l8_w) / 1e6, (l8_f + l8_w) / 192890880)
print(txt)
if "--profiles" in sys.argv:
    body = open(os.path.join(out, "summary.txt")).read()
    open(os.path.join(ROOT, "profiles", "r04_hbm_traffic.txt"), "w").write(txt + body)
    j["source"] = "profiles/r04_hbm_traffic.txt (" + j["source"] + ")"
    json.dump(j, open(os.path.join(ROOT, "profiles", "r04_hbm_traffic.json"), "w"), indent=1)
