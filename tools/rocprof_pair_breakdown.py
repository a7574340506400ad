#!/usr/bin/env python3
"""Per-kernel breakdown of ONE steady-state stereo pair from a rocprofv3 kernel trace of
bench.py: the window between two consecutive normalize_pair dispatches (a forward's first launch) of the timed region
(bench.py's warm-up contains MIOpen's find-mode benchmarking, which would otherwise swamp
the table).  Usage: rocprof_pair_breakdown.py <kernel_trace.csv> [--pair 6] [--phases --encoders --timeline 10]"""
import argparse
import collections
import csv
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--pair", type=int, default=-3, help="index of the forward (normalize_pair dispatch) that opens the window")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--timeline", type=int, default=None, metavar="IT",
                    help="also list every dispatch of GRU iteration IT of the pair (start offset, duration, grid)")
    ap.add_argument("--phases", action="store_true", help="split the pair into the GRU loop (corr build .. last "
                    "iteration) and the rest (upsampling + the next pair's encoders)")
    ap.add_argument("--encoders", action="store_true", help="with --phases: also list every dispatch of the "
                    "up-sampling + encoder phase (start offset, duration, queue)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # A pair = from its input normalisation (normalize_pair_kernel, the forward's first launch) to the next pair's.  Since round 6
    # the loop's prologue starts beside the feature encoder's tail (the correlation build is no longer a clean boundary), so the
    # pair is cut into "encoders" and "loop" at the loop's own first launch: the pack of the hidden states (act_c8_pack_kernel).
    idx = [i for i, r in enumerate(rows) if "normalize_pair" in r["Kernel_Name"]]
    lo, hi = idx[a.pair], idx[a.pair + 1]
    win = rows[lo:hi]
    cut = next((i for i, r in enumerate(win) if "act_c8_pack" in r["Kernel_Name"]), len(win))
    enc, loop = win[:cut], win[cut:]

    def show(r, t0):
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))[:64]
        print("%9.1f %9.1f %9.1f  q%-3s %-64s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                         (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                         (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), n,
                                         r.get("Grid_Size_X", r.get("Grid_Size", ""))))

    if a.timeline is not None:
        lk = [i for i, r in enumerate(loop) if "lookup" in r["Kernel_Name"] or "corr_feat" in r["Kernel_Name"] or "motion_front" in r["Kernel_Name"]]
        it = loop[lk[a.timeline]:lk[a.timeline + 1]]
        t0 = int(it[0]["Start_Timestamp"])
        print("# dispatches of GRU iteration %d (start offset us, duration us, end offset us, queue, kernel, grid threads); two streams overlap" % a.timeline)
        for r in it:
            show(r, t0)
        print("# iteration wall: %.1f us" % ((int(it[-1]["End_Timestamp"]) - t0) / 1e3))
    if a.phases:
        print("# pair wall (normalisation to the next pair's normalisation): %.2f ms"
              % ((int(rows[hi]["Start_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6))
        table(loop, a.top, "GRU loop (from the first pack of the hidden states: prologue + 32 iterations + mask head + up-sampling; the "
                           "feature encoder's tail and the correlation build run beside its prologue)")
        table(enc, a.top, "input normalisation + encoders up to the loop's first launch")
        if a.encoders:
            t0 = int(win[0]["Start_Timestamp"])
            print("# dispatches from the pair's first launch to 12 launches past the loop's first (start offset us, duration us, end offset us, queue, kernel, grid)")
            for r in win[:cut + 12]:
                show(r, t0)
    else:
        table(win, a.top, "one steady-state pair")


def table(win, top, title):
    t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
    for r in win:
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Kernel_Name"]))
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
        if m:
            f = re.findall(r"(CUDAFunctor_\w+|launch_\w+|FillFunctor|MulFunctor|tanh_kernel\w*|copy_kernel\w*)", r["Kernel_Name"])
            n = "at::" + m.group(1) + ("[" + ",".join(sorted(set(f))) + "]" if f else "")
        if len(n) > 68:
            n = re.sub(r"<.*", "<...>", n)[:68]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = agg[n]
        k[0] += 1
        k[1] += d
        k[2] = min(k[2], d)
        k[3] = max(k[3], d)
    tot = sum(v[1] for v in agg.values())
    print("# %s: wall %.2f ms, kernel time %.2f ms, %d dispatches" % (title, (t1 - t0) / 1e6, tot / 1e6, len(win)))
    print("%-68s %6s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, (c, d, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-68s %6d %11.1f %9.2f %9.2f %9.2f %6.2f" % (n, c, d / 1e3, d / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * d / tot))


if __name__ == "__main__":
    main()
