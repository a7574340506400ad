"""Fixed (K-independent) cost of a conv_c8 launch with the ConvGRU gate epilogue: time vs input channels, 256 outputs at 184x312
(tile shape 1) and 128 outputs (shape 2), against the plain epilogue writing C8S only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c8_check import gtime
from dkt_stereo_amd import conv_c8 as c8
torch.manual_seed(0)
H, W = 184, 312
with torch.no_grad():
    h = torch.tanh(torch.randn(1, 128, H, W, device="cuda:0"))
    cz, cr, cq = (torch.randn(1, 128, H, W, device="cuda:0") for _ in range(3))
    rh = c8.ActC8(1, 128, H, W, "cuda:0")
    z = torch.rand(1, 128, H, W, device="cuda:0")
    hn = torch.empty_like(h)
    hc = c8.ActC8(1, 128, H, W, "cuda:0")
    for cin in (16, 48, 128, 256, 384):
        a = c8.pack(torch.randn(1, cin, H, W, device="cuda:0"))
        zr = torch.nn.Conv2d(cin, 256, 3, padding=1).cuda()
        q = torch.nn.Conv2d(cin, 128, 3, padding=1).cuda()
        o256, o128 = c8.ActC8(1, 256, H, W, "cuda:0"), c8.ActC8(1, 128, H, W, "cuda:0")
        print("cin %3d (%2d chunks): z|r gates cfg1 %.1f us | plain->C8S cfg1 %.1f | q update cfg2 %.1f | plain->C8S cfg2 %.1f" % (
            cin, (cin + 15) // 16,
            gtime(lambda: c8.gate_zr([a], zr, cz, cr, h, rh_c8=rh, cfg=1), 5, 5),
            gtime(lambda: c8.conv2d_c8([a], zr, out_c8=o256, cfg=1), 5, 5),
            gtime(lambda: c8.gate_out([a], q, cq, z, h, hn, out_c8=hc, cfg=2), 5, 5),
            gtime(lambda: c8.conv2d_c8([a], q, out_c8=o128, cfg=2), 5, 5)), flush=True)
