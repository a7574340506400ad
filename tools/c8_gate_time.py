import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from c8_check import gtime
from dkt_stereo_amd import conv, conv_c8 as c8
torch.manual_seed(0)
DEV="cuda:0"
with torch.no_grad():
    B, H, W = 1, 184, 312
    h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    x1, x2 = torch.randn(B, 128, H, W, device=DEV), torch.randn(B, 128, H, W, device=DEV)
    cz, cr, cq = (torch.randn(B, 128, H, W, device=DEV) for _ in range(3))
    zr = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
    ql = torch.nn.Conv2d(384, 128, 3, padding=1).to(DEV)
    ah, a1, a2 = c8.pack(h), c8.pack(x1), c8.pack(x2)
    rh_c8 = c8.ActC8(B, 128, H, W, DEV)
    z = c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=rh_c8, cfg=1)
    hn = torch.empty_like(h); hn_c8 = c8.ActC8(B, 128, H, W, DEV)
    print("zr: " + " ".join("cfg%d %.1f" % (c, gtime(lambda: c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=rh_c8, cfg=c), 5, 4)) for c in (1, 2, 5, 6)))
    print("q : " + " ".join("cfg%d %.1f" % (c, gtime(lambda: c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=c), 5, 4)) for c in (2, 3, 6)))
    h4, cz4, cr4, cq4 = (c8.to_c4(t) for t in (h, cz, cr, cq))
    z4 = c8.gate_zr([ah, a1, a2], zr, cz4, cr4, h4, rh_c8=rh_c8, cfg=1, f32_c4=True)
    hn4 = torch.empty_like(h4)
    c8.gate_out([rh_c8, a1, a2], ql, cq4, z4, h4, hn4, out_c8=hn_c8, cfg=2, f32_c4=True)
    z0, rh0 = conv.conv2d_gate_zr([h, x1, x2], zr, cz, cr, h)
    hn0 = conv.conv2d_gate_out([rh0, x1, x2], ql, cq, z0, h)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("C4 gates: z %.2e rh %.2e h' %.2e (c8 %.2e)" % (rel(c8.from_c4(z4), z0), rel(c8.unpack(rh_c8), rh0), rel(c8.from_c4(hn4), hn0), rel(c8.unpack(hn_c8), hn0)))
    print("zr C4: " + " ".join("cfg%d %.1f" % (c, gtime(lambda: c8.gate_zr([ah, a1, a2], zr, cz4, cr4, h4, rh_c8=rh_c8, cfg=c, f32_c4=True), 5, 4)) for c in (1, 2, 5, 6)))
    print("q  C4: " + " ".join("cfg%d %.1f" % (c, gtime(lambda: c8.gate_out([rh_c8, a1, a2], ql, cq4, z4, h4, hn4, out_c8=hn_c8, cfg=c, f32_c4=True), 5, 4)) for c in (2, 3, 6)))
    print("old zr %.1f q %.1f" % (gtime(lambda: conv.conv2d_gate_zr([h, x1, x2], zr, cz, cr, h), 5, 4), gtime(lambda: conv.conv2d_gate_out([h, x1, x2], ql, cq, z, h), 5, 4)))
