#!/usr/bin/env python3
"""traffic.json (and, with --profiles, profiles/r06_hbm_traffic[_b<B>].{txt,json}) from a tools/pmc/run_pmc.sh output
directory: bytes per launch = FETCH_SIZE / (calibrated fraction of the bytes read) + WRITE_SIZE / (calibrated fraction written),
averaged over the loop's own dispatches (the last 32 fused ConvGRU steps, the last 31 motion fronts of the probe process)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
pmc = json.load(open(os.path.join(out, "pmc.json")))
rows, fac = pmc["rows"], pmc["factors"]
KB = 1024
ff = fac.get("calib_read16.fetch", 0.5)          # FETCH_SIZE reports this fraction of the bytes read (0.500 on gfx950)
wf = fac.get("calib_write4.write", 1.0)


def tail(prefix, n):
    k = [k for k in rows if k.startswith(prefix)][0]
    f, w = rows[k]["fetch_kb"][-n:], rows[k]["write_kb"][-n:]
    return KB * sum(f) / len(f) / ff, KB * sum(w) / len(w) / wf


g_f, g_w = tail("gru_c8_kernel", 32)
m_f, m_w = tail("motion_front_kernel", 31)
o_f, o_w = tail("corr1d_lookup_skew_kernel", 3)
px, px32 = 184 * 312 * B, 46 * 78 * B
# compulsory bytes of one fused ConvGRU step: per pixel of gru08 the C8S operands h, x1, x2 read by z|r (1536 B) and x1, x2, r*h
# by q (1536), the context terms cz, cr, cq (1536), the state h once (512), r*h written (512), h' as fp32 and C8S (1024);
# the riding gru32 at 46x78 has ONE x tensor (1024 + 1024 + 1536 + 512 + 512 + 1024); weights 5.3 + 3.5 MB once per launch
comp = px * 6656 + px32 * 5632 + 8.8e6
alg_front = px * 764
alg_op = px * 308
j = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch of the loop's own launches (tools/pmc/run_pmc.sh); "
              "reads = FETCH_SIZE / %.3f, writes = WRITE_SIZE / %.3f from 1 GiB known-traffic streams in the same passes" % (ff, wf),
    "shape": [736, 1248, B], "gru_rider_hw": [46, 78],
    "fetch_fraction": ff, "write_fraction": wf,
    "gru_bytes": int(g_f + g_w), "gru_fetch_bytes": int(g_f), "gru_write_bytes": int(g_w), "gru_compulsory_bytes": int(comp),
    "motion_front_bytes": int(m_f + m_w), "motion_front_algorithmic_bytes": int(alg_front),
    "lookup_operator_bytes": int(o_f + o_w), "lookup_operator_algorithmic_bytes": int(alg_op),
}
json.dump(j, open(os.path.join(out, "traffic.json"), "w"), indent=1)
txt = """# HBM-side traffic of the kernels bench.py reports (MI355X, round 6, batch %d): bash tools/pmc/run_pmc.sh %d
#   fused ConvGRU launch (gru_c8_kernel, gru08 184x312 + gru32 46x78 -- the loop's own dispatches): %.1f MB read + %.1f MB written
#       = %.1f MB per launch vs %.1f MB compulsory = %.2fx
#   motion front (coordinate update + lookup + convc1 + 7x7 stem -> C8S): %.1f MB vs %.1f MB algorithmic = %.2fx
#   reference-visible lookup operator (corr1d_lookup_skew_kernel): %.1f MB vs %.1f MB algorithmic = %.2fx
""" % (B, B, g_f / 1e6, g_w / 1e6, (g_f + g_w) / 1e6, comp / 1e6, (g_f + g_w) / comp,
       (m_f + m_w) / 1e6, alg_front / 1e6, (m_f + m_w) / alg_front, (o_f + o_w) / 1e6, alg_op / 1e6, (o_f + o_w) / alg_op)
print(txt)
if "--profiles" in sys.argv:
    sfx = "" if B == 1 else "_b%d" % B
    body = open(os.path.join(out, "summary_tail.txt")).read()
    open(os.path.join(ROOT, "profiles", "r06_hbm_traffic%s.txt" % sfx), "w").write(txt + body)
    j["source"] = "profiles/r06_hbm_traffic%s.txt (%s)" % (sfx, j["source"])
    json.dump(j, open(os.path.join(ROOT, "profiles", "r06_hbm_traffic%s.json" % sfx), "w"), indent=1)
