#!/usr/bin/env python3
"""Runs one hot-path kernel in isolation a few times (for rocprofv3 --pmc / --kernel-trace).
usage: prof_conv.py [zr|zr_gate|q|enc|lookup|build] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "zr"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
from dkt_stereo_amd import conv  # noqa: E402
from dkt_stereo_amd.corr import CorrBlock1D  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
with torch.no_grad():
    if which in ("zr", "q"):
        cout = 256 if which == "zr" else 128
        conv.set_backend(os.environ.get("DKT_CONV", "f16x3"))
        layer = torch.nn.Conv2d(384, cout, 3, padding=1).to(dev)
        xs = [torch.randn(1, 128, 184, 312, device=dev) for _ in range(3)]
        for _ in range(reps):
            conv.conv2d(xs, layer)
    elif which == "zr_gate":      # the dominant kernel of bench.py: gru08 z|r convolution with the gate epilogue
        conv.set_backend("f16x3")
        layer = torch.nn.Conv2d(384, 256, 3, padding=1).to(dev)
        xs = [torch.tanh(torch.randn(1, 128, 184, 312, device=dev))] + [torch.randn(1, 128, 184, 312, device=dev) for _ in range(2)]
        cz, cr = (torch.randn(1, 128, 184, 312, device=dev) for _ in range(2))
        for _ in range(reps):
            conv.conv2d_gate_zr(xs, layer, cz, cr, xs[0])
    elif which == "small":        # a latency-bound small layer: motion encoder convc2, 64->64 3x3 at 184x312
        conv.set_backend("f16x3")
        layer = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev)
        x = torch.randn(1, 64, 184, 312, device=dev)
        for _ in range(reps):
            conv.conv2d(x, layer, relu=True)
    elif which == "enc":          # narrow encoder layer: 64->64 3x3 at 368x624
        conv.set_backend("f16x3")
        layer = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev)
        x = torch.randn(1, 64, 368, 624, device=dev)
        for _ in range(reps):
            conv.conv2d(x, layer)
    elif which == "lookup8":      # cfg4's per-GPU share: batch 8, smooth disparity (what the GRU loop produces)
        f1, f2 = (torch.randn(8, 256, 184, 312, device=dev) for _ in range(2))
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        coords = torch.zeros(8, 2, 184, 312, device=dev)
        xs = torch.arange(312, device=dev).float().view(1, 1, 312)
        coords[:, 0] = xs - (10.0 + 30.0 * xs / 312) - 0.3 * torch.rand(8, 184, 312, device=dev)
        for _ in range(reps):
            blk(coords)
    elif which in ("lookup", "build"):
        f1, f2 = (torch.randn(1, 256, 184, 312, device=dev) for _ in range(2))
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        coords = torch.zeros(1, 2, 184, 312, device=dev)
        coords[:, 0] = torch.arange(312, device=dev).float().view(1, 1, 312) - 60 * torch.rand(1, 184, 312, device=dev)
        for _ in range(reps):
            if which == "lookup":
                blk(coords)
            else:
                CorrBlock1D(f1, f2, num_levels=4, radius=4)
    torch.cuda.synchronize()
print("done", which)
