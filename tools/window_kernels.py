#!/usr/bin/env python3
"""Stand-alone launch time of every convolution of the window between two fused-GRU launches (loop_c8.C8Loop.unit), per tile
shape (conv_c8.hip c8_dispatch cfg) and batch size: which shapes the per-batch defaults of loop_c8 should be.
    python tools/window_kernels.py [batches...]   (default 1 8)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_kernels import timeit  # noqa: E402
from dkt_stereo_amd import conv_c8 as c8  # noqa: E402
from dkt_stereo_amd.update import _leading_outputs  # noqa: E402

DEV = "cuda:0"
H, W = 184, 312
mk = lambda cin, cout: torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)


@torch.no_grad()
def run(B):
    torch.manual_seed(0)
    print("\n## batch %d (us per launch; us per pair; algorithmic TFLOP/s; frac of the 2.5 PF fp16 peak)" % B)

    def rep(name, fn, gflop):
        try:
            us = timeit(fn, n=max(10, 60 // B))
        except Exception as e:          # shape not covered by this tile shape
            print("%-58s unsupported (%s)" % (name, str(e)[:40]))
            return
        tf = gflop * B / us * 1e3                       # GFLOP / us = PFLOP/s
        print("%-58s %9.1f %8.1f %7.1f %6.3f" % (name, us, us / B, tf, tf / 2500.0), flush=True)

    acts = lambda n, h, w, c=128: [c8.pack(torch.randn(B, c, h, w, device=DEV)) for _ in range(n)]
    hc = acts(1, H, W)[0]
    h1, h2 = mk(128, 256), mk(256, 2)
    tgt = torch.zeros(B, 1, H, W, device=DEV)
    for cfg in (2, 1):
        rep("flow head conv1 128->256 + folded conv2 + finish (cfg%d)" % cfg,
            lambda: c8.head([hc], h1, _leading_outputs(h2, 1), tgt, cfg=cfg), 2e-9 * 9 * 128 * 256 * H * W)
    cor, flo = acts(2, H, W, 64)
    cf, mf = c8.ActC8(B, 128, H, W, DEV), c8.ActC8(B, 128, H, W, DEV, tail=2)
    c2, f2, enc = mk(64, 64), mk(64, 64), mk(128, 126)
    flow = torch.randn(B, 2, H, W, device=DEV)
    for cfg in (4, 3):
        def pair():
            d0 = c8.desc([cor], c2, relu=True, out_c8=cf, out_c8_ch0=0)
            d1 = c8.desc([flo], f2, relu=True, out_c8=cf, out_c8_ch0=64)
            c8.launch_pair(d0, d1, flow, cfg)
        rep("convc2 | convf2 64->64 x2 (cfg%d)" % cfg, pair, 2 * 2e-9 * 9 * 64 * 64 * H * W)
    for cfg in (3, 4, 2):
        rep("encoder.conv 128->126 + tail (cfg%d)" % cfg, lambda: c8.conv2d_c8([cf], enc, relu=True, out_c8=mf, tail=flow, cfg=cfg),
            2e-9 * 9 * 128 * 126 * H * W)
    h, w = 92, 156
    a = acts(3, h, w)
    hs = torch.tanh(torch.randn(B, 128, h, w, device=DEV))
    g0, g1, g2 = (torch.randn(B, 128, h, w, device=DEV) for _ in range(3))
    r2, hh2 = c8.ActC8(B, 128, h, w, DEV), c8.ActC8(B, 128, h, w, DEV)
    lz, lq = mk(384, 256), mk(384, 128)
    zz = c8.gate_zr(a, lz, g0, g1, hs, rh_c8=r2, cfg=4)
    for cfg in (4, 3, 2, 1):
        rep("gru16 z|r 384->256 + gates (cfg%d)" % cfg, lambda: c8.gate_zr(a, lz, g0, g1, hs, rh_c8=r2, cfg=cfg), 2e-9 * 9 * 384 * 256 * h * w)
    for cfg in (4, 3, 2):
        rep("gru16 q 384->128 + update (cfg%d)" % cfg, lambda: c8.gate_out(a, lq, g2, zz, hs, hs, out_c8=hh2, cfg=cfg), 2e-9 * 9 * 384 * 128 * h * w)
    n0, n1, n2 = torch.randn(B, 128, H, W, device=DEV), torch.randn(B, 128, h, w, device=DEV), torch.randn(B, 128, 46, 78, device=DEV)
    p0, u1, u2, p1 = c8.ActC8(B, 128, h, w, DEV), c8.ActC8(B, 128, H, W, DEV), c8.ActC8(B, 128, h, w, DEV), c8.ActC8(B, 128, 46, 78, DEV)
    rep("pool2x(1/4) | interp(1/16 -> 1/8)", lambda: c8.resample_pair_c8(("pool", n0, p0), ("interp", n2, u2)), 0.0)
    rep("interp(1/8 -> 1/4) | pool2x(1/8)", lambda: c8.resample_pair_c8(("interp", n1, u1), ("pool", n1, p1)), 0.0)


if __name__ == "__main__":
    for b in [int(x) for x in sys.argv[1:]] or [1, 8]:
        run(b)
