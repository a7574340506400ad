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
with torch.no_grad():
    for cin, cout, k, H, W, B in [(256, 2, 3, 184, 312, 1), (256, 1, 3, 184, 312, 1), (2, 64, 7, 184, 312, 1), (3, 64, 7, 736, 1248, 1), (3, 64, 7, 736, 1248, 2)]:
        layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(DEV)
        x = torch.randn(B, cin, H, W, device=DEV)
        ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=k // 2)
        got = conv._conv2d_direct(x, layer, False, None)
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        t_d = timeit(lambda: conv._conv2d_direct(x, layer, False, None))
        conv.set_backend("miopen"); t_m = timeit(lambda: conv.conv2d(x, layer)); conv.set_backend("f16x3")
        t_f = timeit(lambda: conv.conv2d([x], layer))   # default path (single-element list is unwrapped)
        print("%d->%d k%d %dx%d B=%d  direct %.1f us  miopen %.1f us  conv2d(default) %.1f us  rel err %.1e" % (cin, cout, k, H, W, B, t_d, t_m, t_f, err), flush=True)
