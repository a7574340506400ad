#!/usr/bin/env python3
"""Numerics study (CPU, oracle only; not a test): why does the error of the two-product convolutions (activations rounded to
fp16) grow LINEARLY with the number of reduced iterations, and would a dithered rounding change that?

The oracle's update-block convolutions are re-run with their 3x3 inputs rounded to fp16 (a) to nearest -- what
dkt_conv_c8_desc.passes = 2 computes -- and (b) stochastically (x + u * ulp, u ~ U(-1/2, 1/2), then to nearest: unbiased, and a
different error in every iteration), and (c) as fp16 hi x hi plus the two cross terms on fp8 (e4m3) operands -- a form gfx950's
f8f6f4 MFMAs would run in 2 product-equivalents; final disparity against the exact forward, per number of iterations.
    python tests/study_rounding.py [HxW] [iters]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _cases  # noqa: E402
import _synth  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402
from dkt_stereo_amd.raft_stereo import BASE_CONFIG, RAFTStereo  # noqa: E402

H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x512").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
model = RAFTStereo()
sd = _synth.torch_state_dict(_synth.shapes_of(model), _cases.E2E_WEIGHT_SEED)
i1, i2 = (torch.from_numpy(t) for t in _synth.image_pair(2, 1, H, W, 40))
real_conv = to._conv
mode = {"m": "exact"}
gen = torch.Generator().manual_seed(1)


def rounded(x):
    if mode["m"] == "rn":
        return x.half().float()
    e = torch.floor(torch.log2(x.abs().clamp_min(2.0 ** -14)))
    ulp = torch.pow(2.0, e - 10)
    u = torch.rand(x.shape, generator=gen) - 0.5
    return (x + u * ulp).half().float()


def q8(t):
    """fp8 (e4m3) rounding with a per-tensor power-of-two scale (the hardware's MX formats carry one per 32 elements: this is
    the pessimistic end)."""
    m = float(t.abs().max())
    if m == 0.0:
        return t
    s = 2.0 ** torch.floor(torch.log2(torch.tensor(448.0 / m))).item()
    return (t * s).to(torch.float8_e4m3fn).float() / s


def conv(sd_, name, x, stride=1):
    w = sd_[name + ".weight"]
    if mode["m"] != "exact" and name.startswith("update_block.") and w.shape[2] == 3 and w.shape[1] >= 64:
        if mode["m"] == "f8x":
            # hi x hi on fp16, the two cross terms on fp8 operands (half the MFMA time each on gfx950): 2 product-equivalents
            import torch.nn.functional as F
            xh, wh = x.half().float(), w.half().float()
            xl, wl = x - xh, w - wh
            pad = (w.shape[2] // 2, w.shape[3] // 2)
            y = F.conv2d(xh, wh, sd_[name + ".bias"], stride=stride, padding=pad)
            y = y + F.conv2d(q8(xl), q8(wh), None, stride=stride, padding=pad) + F.conv2d(q8(xh), q8(wl), None, stride=stride, padding=pad)
            return y
        x = rounded(x)
    return real_conv(sd_, name, x, stride)


to._conv = conv
with torch.no_grad():
    res = {}
    for m in ("exact", "rn", "sr", "f8x"):
        mode["m"] = m
        res[m] = [to.raft_stereo_forward(sd, dict(BASE_CONFIG), i1, i2, k)[1] for k in (iters // 4, iters // 2, iters)]
print("# %dx%d, final disparity: max-abs / mean-abs against the exact forward (oracle, CPU)" % (H, W))
print("%-10s %28s %28s %32s" % ("iterations", "fp16 to nearest (2 products)", "fp16 dithered", "fp16 hi*hi + fp8 cross terms"))
for j, k in enumerate((iters // 4, iters // 2, iters)):
    d = [(res[m][j] - res["exact"][j]).abs() for m in ("rn", "sr", "f8x")]
    print("%-10d %14.3e / %9.3e %16.3e / %9.3e %18.3e / %9.3e" % (k, float(d[0].max()), float(d[0].mean()), float(d[1].max()), float(d[1].mean()),
                                                                float(d[2].max()), float(d[2].mean())))
