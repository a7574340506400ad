import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dkt_stereo_amd import conv
DEV = "cuda:0"
torch.manual_seed(0)
cases = [([128], 128, 3, 8, 12, 1), ([128, 128], 128, 3, 4, 6, 1), ([128, 128, 128], 256, 3, 8, 12, 1), ([128], 128, 3, 2, 3, 1),
         ([128, 128], 256, 3, 2, 3, 1), ([36], 64, 1, 8, 12, 1), ([64], 64, 3, 8, 12, 1), ([64, 64], 126, 3, 8, 12, 1),
         ([128], 256, 3, 8, 12, 2), ([256], 2, 3, 8, 12, 1), ([128], 128, 3, 1, 2, 1), ([128, 128], 128, 3, 1, 2, 1), ([40], 64, 3, 9, 70, 3)]
with torch.no_grad():
    for chans, cout, k, H, W, B in cases:
        layer = torch.nn.Conv2d(sum(chans), cout, k, padding=k // 2).to(DEV)
        xs = [torch.randn(B, c, H, W, device=DEV) for c in chans]
        ref = F.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), padding=k // 2)
        got = conv.conv2d(xs if len(xs) > 1 else xs[0], layer)
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        print(chans, cout, k, H, W, B, "rel err %.2e" % err, "NaN" if got.isnan().any() else "", flush=True)
    print("-- gates")
    for H, W, B, nx in [(8, 12, 1, 2), (4, 6, 1, 2), (2, 3, 1, 1), (46, 78, 1, 2), (184, 312, 1, 2)]:
        ch = 128
        xs = [torch.tanh(torch.randn(B, ch, H, W, device=DEV))] + [torch.randn(B, 128, H, W, device=DEV) for _ in range(nx)]
        zr = torch.nn.Conv2d(128 * (nx + 1), 256, 3, padding=1).to(DEV)
        ql = torch.nn.Conv2d(128 * (nx + 1), 128, 3, padding=1).to(DEV)
        cz, cr, cq = (torch.randn(B, ch, H, W, device=DEV) for _ in range(3))
        h = xs[0]
        v = F.conv2d(torch.cat(xs, 1).double(), zr.weight.double(), zr.bias.double(), padding=1)
        z_ref = torch.sigmoid(v[:, :ch] + cz.double()); r_ref = torch.sigmoid(v[:, ch:] + cr.double())
        z, rh = conv.conv2d_gate_zr(xs, zr, cz, cr, h)
        e1 = (z.double() - z_ref).abs().max().item(); e2 = (rh.double() - r_ref * h.double()).abs().max().item()
        xq = [rh] + xs[1:]
        vq = F.conv2d(torch.cat(xq, 1).double(), ql.weight.double(), ql.bias.double(), padding=1)
        h_ref = (1 - z.double()) * h.double() + z.double() * torch.tanh(vq + cq.double())
        hn = conv.conv2d_gate_out(xq, ql, cq, z, h)
        e3 = (hn.double() - h_ref).abs().max().item()
        print(H, W, B, nx, "z %.2e rh %.2e h %.2e" % (e1, e2, e3), flush=True)
