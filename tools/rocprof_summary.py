#!/usr/bin/env python3
"""Condenses a rocprofv3 kernel trace (ROCm 7.2: rocpd sqlite .db or *_kernel_trace.csv) into a
short per-kernel table: calls, total, average, min, max duration (microseconds) and share.
Usage: rocprof_summary.py <results.db | kernel_trace.csv> [--top N] [--match substr] [--skip-first K]"""
import argparse
import csv
import re
import sqlite3


def short(name, width):
    name = re.sub(r"\(.*", "", name)                  # drop argument lists
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"<.*", "<...>", name) if len(name) > width else name
    return name[:width]


def load(path):
    if path.endswith(".csv"):
        rows = []
        for r in csv.DictReader(open(path)):
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        return rows
    c = sqlite3.connect(path)
    return c.execute("select name, (end - start) from kernels").fetchall()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--match", default=None)
    ap.add_argument("--width", type=int, default=72)
    a = ap.parse_args()
    rows = load(a.trace)
    agg = {}
    for name, dur in rows:
        k = agg.setdefault(name, [0, 0, 1 << 62, 0])
        k[0] += 1
        k[1] += dur
        k[2] = min(k[2], dur)
        k[3] = max(k[3], dur)
    total = sum(v[1] for v in agg.values()) or 1
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if a.match:
        items = [kv for kv in items if a.match in kv[0]]
    print("%-*s %8s %12s %10s %10s %10s %6s" % (a.width, "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, (n, tot, mn, mx) in items[:a.top]:
        print("%-*s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (a.width, short(name, a.width), n, tot / 1e3,
                                                             tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print("# %d kernels, %d dispatches, %.1f ms of kernel time" % (len(agg), len(rows), total / 1e6))


if __name__ == "__main__":
    main()
