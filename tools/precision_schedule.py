#!/usr/bin/env python3
"""Precision schedules of the refinement loop (loop_c8.SCHEDULE): the first k1 iterations at ONE fp16 MFMA
product per block, the next k2 at TWO, the rest at the fp32-class THREE -- final disparity against the reference's own fixtures
(tests/golden/raft_e2e.npz: max-abs and EPE on the fixture's pixel grid) for every e2e fixture and for BASELINE cfg4's per-GPU
share (8 pairs per launch, pair 0 = the 736x1248 fixture pair), with the throughput of each schedule at the benchmark shape.
The reference's switch of this kind is `mixed_precision` (raft_stereo.py:95,156; tools/evaluate_stereo.py:380).
    python tools/precision_schedule.py [--quick]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases  # noqa: E402
import _synth  # noqa: E402
from dkt_stereo_amd.raft_stereo import RAFTStereo  # noqa: E402

dev = "cuda:0"
quick = "--quick" in sys.argv
G = np.load(os.path.join(ROOT, "tests", "golden", "raft_e2e.npz"))


def model_():
    m = RAFTStereo()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), _cases.E2E_WEIGHT_SEED), strict=True)
    return m.to(dev).eval()


def schedules(iters):
    ks = [k for k in range(0, 33, 4) if k <= iters] if iters >= 8 else list(range(0, iters + 1))
    out = [("fp32-class", None)]
    out += [("%2d x 2 passes, then 3" % k, (0, k)) for k in ks if k]
    out += [("%2d x 1 pass , then 3" % k, (k, 0)) for k in ks if k]
    if iters >= 32:
        out += [("%d x 1, %d x 2, then 3" % (a, b), (a, b)) for a, b in ((8, 8), (8, 16), (16, 8), (12, 12), (16, 12), (20, 8), (24, 4))]
    return out


@torch.no_grad()
def run(name, c, model, timing):
    B = c.get("batch", 1)
    pairs = [_synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"])]
    pairs += [_synth.image_pair(2000 + j, 1, c["H"], c["W"], 12 if j % 2 else 40) for j in range(1, B)]
    i1 = torch.cat([torch.from_numpy(p[0]) for p in pairs]).to(dev)
    i2 = torch.cat([torch.from_numpy(p[1]) for p in pairs]).to(dev)
    s = int(G[c["fixture"] + "/stride"])
    ref = G[c["fixture"] + "/flow_up"]
    print("\n## %s  (%dx%d, %d iterations, batch %d; fixture %s)" % (name, c["H"], c["W"], c["iters"], B, c["fixture"]))
    print("%-26s %12s %12s %14s %s" % ("schedule (iterations)", "max-abs", "EPE", "vs fp32-class", "pairs/s" if timing else ""))
    base = None
    for label, sch in schedules(c["iters"]):
        model.precision_schedule = sch
        for _ in range(3 if timing else 1):
            _, up = model(i1, i2, iters=c["iters"], test_mode=True)
        d = np.abs(up[:1, :, ::s, ::s].cpu().numpy() - ref)
        if base is None:
            base = up.clone()
        rate = ""
        if timing:
            model.check_finite = False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3 if B > 1 else 8
            for _ in range(n):
                model(i1, i2, iters=c["iters"], test_mode=True)
            torch.cuda.synchronize()
            rate = "%.1f" % (B * n / (time.perf_counter() - t0))
            model.check_finite = True
        print("%-26s %12.3e %12.3e %14.3e %s  %s" % (label, d.max(), d.mean(), float((up - base).abs().max()), rate,
                                                    "<= 1e-3" if d.max() <= 1e-3 else ""), flush=True)
    model.precision_schedule = None


cases = []
for name, c in _cases.E2E_CASES.items():
    cases.append((name, dict(c, fixture=name), name == "736x1248_it32"))
big = _cases.E2E_CASES["736x1248_it32"]
cases.append(("cfg4 share: 8 pairs per launch", dict(big, fixture="736x1248_it32", batch=8), True))
if quick:
    cases = [cases[1], cases[4]]
print("# final disparity vs the reference's fixtures under precision schedules of the refinement loop (MI355X; encoders, correlation")
print("# volume, lookup and up-sampling always fp32-class; 'vs fp32-class' = max-abs against this library's own default path)")
last_shape = None
for name, c, timing in cases:
    if (c["H"], c["W"], c.get("batch", 1)) != last_shape:
        model = model_()
        last_shape = (c["H"], c["W"], c.get("batch", 1))
    run(name, c, model, timing)
