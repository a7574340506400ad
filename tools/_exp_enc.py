"""Encoder-layer conv timings per tile shape (DKT_CONV_CFG read once per process: run per cfg)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dkt_stereo_amd import conv
DEV = "cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
cases = [(256, 256, 46, 78, 1), (256, 128, 46, 78, 1), (384, 256, 92, 156, 1), (384, 128, 92, 156, 1), (384, 256, 184, 312, 1), (384, 128, 184, 312, 1), (64, 64, 368, 624, 1), (64, 64, 368, 624, 2), (64, 64, 184, 312, 1), (96, 96, 184, 312, 1), (96, 96, 184, 312, 2),
         (128, 128, 184, 312, 1), (128, 128, 184, 312, 2), (128, 256, 184, 312, 1), (128, 128, 92, 156, 1)]
with torch.no_grad():
    for cin, cout, H, W, B in cases:
        layer = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
        x = torch.randn(B, cin, H, W, device=DEV)
        if cin in (256, 384) and H < 200:
            x = [torch.randn(B, 128, H, W, device=DEV) for _ in range(cin // 128)]
        be = os.environ.get("BACKEND", "f16x3")
        conv.set_backend(be)
        us = timeit(lambda: conv.conv2d(x, layer, relu=True))
        fl = 2.0 * B * H * W * cin * 9 * cout
        print("cfg=%s %s %3d->%3d %dx%d B=%d  %8.1f us  %6.1f TF(fp32-equiv)" % (os.environ.get("DKT_CONV_CFG", "auto"), be, cin, cout, H, W, B, us, fl / us / 1e6), flush=True)
