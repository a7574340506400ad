#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB as reported) from the two counter-collection CSVs of run_pmc.sh, the
calibration factors of the known-traffic kernels, and profiles/r02_hbm_traffic.json for bench.py."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                vals[k][c].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
GiB = 1024 ** 3
known = {"calib_copy4": (GiB, GiB), "calib_copy16": (GiB, GiB), "calib_read4": (GiB, 0), "calib_read16": (GiB, 0), "calib_write4": (0, GiB)}
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per dispatch (KB as reported), MI355X round 2")
print("%-62s %14s %14s" % ("kernel (dispatch order)", "FETCH_SIZE KB", "WRITE_SIZE KB"))
factors = {}
rows = {}
for k in sorted(vals):
    f = [v for _, v in sorted(vals[k]["FETCH_SIZE"])]
    w = [v for _, v in sorted(vals[k]["WRITE_SIZE"])]
    rows[k] = (f, w)
    for i in range(max(len(f), len(w))):
        print("%-62s %14.1f %14.1f" % (k[:62], f[i] if i < len(f) else float("nan"), w[i] if i < len(w) else float("nan")))
print()
print("# calibration: reported / true bytes (1 GiB streams, 4x the Infinity Cache)")
for k, (rb, wb) in known.items():
    f, w = rows.get(k, ([], []))
    if rb and f:
        factors[k + ".fetch"] = sum(f) / len(f) * 1024 / rb
        print("%-16s FETCH_SIZE reports %.3f of the bytes read" % (k, factors[k + ".fetch"]))
    if wb and w:
        factors[k + ".write"] = sum(w) / len(w) * 1024 / wb
        print("%-16s WRITE_SIZE reports %.3f of the bytes written" % (k, factors[k + ".write"]))
json.dump({"factors": factors, "rows": {k: {"fetch_kb": v[0], "write_kb": v[1]} for k, v in rows.items()}},
          open(os.path.join(out, "pmc.json"), "w"), indent=1)
