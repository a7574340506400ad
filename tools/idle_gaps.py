#!/usr/bin/env python3
"""Where the device idles inside one stereo pair: from a rocprofv3 kernel trace, the window between two consecutive
normalize_pair dispatches (a forward's first launch); every interval of at least --min microseconds in which NO queue runs a kernel, with the kernels on
either side.  Used to compare the product's default mode (check_finite = True: one host synchronisation per pair) with the
unchecked mode bench.py's timed region runs.   idle_gaps.py <kernel_trace.csv> [--pair -3] [--min 10]"""
import argparse
import csv
import re


def name(r):
    return re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--pair", type=int, default=-3)
    ap.add_argument("--min", type=float, default=10.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "normalize_pair" in r["Kernel_Name"]]
    lo, hi = idx[a.pair], idx[a.pair + 1]
    win = rows[lo:hi]
    t0 = int(win[0]["Start_Timestamp"])
    busy_end, last = int(win[0]["End_Timestamp"]), win[0]
    idle, gaps = 0, []
    for r in win[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > busy_end:
            idle += s - busy_end
            if (s - busy_end) / 1e3 >= a.min:
                gaps.append(((busy_end - t0) / 1e3, (s - busy_end) / 1e3, name(last), name(r)))
        if e > busy_end:
            busy_end, last = e, r
    wall = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
    print("# pair window %.1f us, device idle %.1f us in total; gaps >= %.0f us:" % (wall, idle / 1e3, a.min))
    for at, d, before, after in gaps:
        print("%10.1f  idle %7.1f us   after %-48s before %s" % (at, d, before, after))


if __name__ == "__main__":
    main()
