"""Weights-stationary 64 -> 64 convolution (csrc/conv_ws.h) against an fp64 convolution, and its launch time.
Run twice (DKT_CONV_WS=1 / 0) for the A/B against the streaming kernel:  python tools/conv_ws_check.py [--time]"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dkt_stereo_amd import conv  # noqa: E402
from dkt_stereo_amd.extractor import instance_norm_params  # noqa: E402


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def check(B, H, W, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    layer = nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(B, 64, H, W, device="cuda", generator=g)
    res = torch.randn(B, 64, H, W, device="cuda", generator=g).relu()
    wd, bd = layer.weight.detach().double(), layer.bias.detach().double()
    with torch.no_grad():
        ref = F.conv2d(x.double(), wd, bd, padding=1)
        y = conv.conv2d(x, layer, relu=True)
        e_relu = rel(y, ref.relu())
        y = conv.conv2d_fused(x, layer, relu=True, residual=res)
        e_join = rel(y, (res.double() + ref.relu()).relu())
        norm = nn.InstanceNorm2d(64)
        p = instance_norm_params(norm, x)
        xn = F.instance_norm(x.double()).relu()
        refn = F.conv2d(xn, wd, bd, padding=1)
        y, st = conv.conv2d_stats(x, layer, in_norm=p)
        e_norm = rel(y, refn)
        p2 = instance_norm_params(norm, y, st).view(B, 64, 2).double()
        mean = refn.mean(dim=(2, 3))
        var = refn.var(dim=(2, 3), unbiased=False)
        e_mean = float((p2[..., 0] - mean).abs().max() / mean.abs().max())
        e_istd = float((p2[..., 1] - (var + 1e-5).rsqrt()).abs().max() / (var + 1e-5).rsqrt().abs().max())
    print("B=%d %dx%d  relu %.2e  join %.2e  in_norm %.2e  stats mean %.2e 1/std %.2e" % (B, H, W, e_relu, e_join, e_norm, e_mean, e_istd), flush=True)
    return max(e_relu, e_join, e_norm, e_mean, e_istd)


def timeit(B, H, W, n=20):
    layer = nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(B, 64, H, W, device="cuda")
    res = torch.randn(B, 64, H, W, device="cuda").relu()
    norm = nn.InstanceNorm2d(64)
    with torch.no_grad():
        p = instance_norm_params(norm, x)
        forms = {"relu": lambda: conv.conv2d(x, layer, relu=True),
                 "join": lambda: conv.conv2d_fused(x, layer, relu=True, residual=res),
                 "in_norm+stats": lambda: conv.conv2d_stats(x, layer, in_norm=p)}
        for name, f in forms.items():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            gf = 2.0 * B * H * W * 64 * 64 * 9 / 1e9
            print("B=%d %dx%d %-14s %8.1f us per launch  (%.1f us per image, %.0f TF algorithmic = %.3f of 2.5 PF)"
                  % (B, H, W, name, us, us / B, gf / us * 1e3, gf / us / 2.5), flush=True)


if __name__ == "__main__":
    print("DKT_CONV_WS =", os.environ.get("DKT_CONV_WS", "(default 1)"))
    worst = 0.0
    for shape in ((1, 736, 1248), (2, 736, 1248), (3, 250, 332), (1, 544, 960)):
        worst = max(worst, check(*shape))
    print("worst", worst)
    if "--time" in sys.argv:
        timeit(1, 736, 1248)
        timeit(2, 736, 1248)
    sys.exit(0 if worst < 5e-6 else 1)
