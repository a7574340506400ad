"""Fixed (K-independent) cost of the C8S convolution: time vs input channels at 184x312."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c8_check import gtime
from dkt_stereo_amd import conv, conv_c8 as c8
torch.manual_seed(0)
with torch.no_grad():
    for cout, cfg in ((256, 1), (128, 2), (64, 3), (64, 4)):
        r = []
        for cin in (16, 64, 128, 256, 384):
            x = torch.randn(1, cin, 184, 312, device="cuda:0")
            a = c8.pack(x)
            layer = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
            oc = c8.ActC8(1, cout, 184, 312, "cuda:0")
            t_nchw = gtime(lambda: c8.conv2d_c8([a], layer, cfg=cfg), 5, 4)
            t_c8 = gtime(lambda: c8.conv2d_c8([a], layer, out_c8=oc, cfg=cfg), 5, 4)
            t_old = gtime(lambda: conv.conv2d(x, layer), 5, 4) if cin >= 64 else float("nan")
            r.append("cin %3d: nchw %.1f c8out %.1f old %.1f" % (cin, t_nchw, t_c8, t_old))
        print("cout %d cfg %d | " % (cout, cfg) + " | ".join(r), flush=True)
