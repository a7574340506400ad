"""Numerics and timing of the C8S convolution (csrc/conv_c8.hip) against the round-2 kernel and an fp64 reference."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dkt_stereo_amd import conv, conv_c8 as c8
DEV = "cuda:0"
torch.manual_seed(0)


def gtime(fn, reps=5, n=6):
    """Median over n separately timed replays of a graph of `reps` launches, after >= 6 warm replays (clock / power state
    settle: the first replays after another kernel mix can be 10-15 % off)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(max(n, 5)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@torch.no_grad()
def main():
    quick = "--quick" in sys.argv
    for (B, H, W, chs, cout) in [(1, 64, 96, [128, 128, 128], 256), (1, 184, 312, [128, 128, 128], 256),
                                  (1, 184, 312, [128, 128, 128], 128), (1, 184, 312, [128], 256), (1, 184, 312, [64, 64], 126),
                                  (1, 92, 156, [128, 128, 128], 256), (1, 46, 78, [128, 128], 256), (2, 50, 70, [64], 64)]:
        xs = [torch.randn(B, c, H, W, device=DEV) for c in chs]
        layer = torch.nn.Conv2d(sum(chs), cout, 3, padding=1).to(DEV)
        ref64 = torch.nn.functional.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), padding=1)
        old = conv.conv2d(xs, layer)
        acts = [c8.pack(x) for x in xs]
        for a, x in zip(acts, xs):
            assert rel(c8.unpack(a), x) < 1e-6
        line = "B%d %dx%d %s->%d: old %.2e" % (B, H, W, chs, cout, rel(old, ref64))
        for cfg in (1, 2, 3, 4, 5, 6):
            y = c8.conv2d_c8(acts, layer, cfg=cfg)
            oc = c8.ActC8(B, cout, H, W, DEV)
            c8.conv2d_c8(acts, layer, out_c8=oc, cfg=cfg)
            line += " | cfg%d %.2e c8out %.2e" % (cfg, rel(y, ref64), rel(c8.unpack(oc), ref64))
            # the border and the padding channels of the C8S output must still be zero
            t = oc.t.clone()
            t[:, :, :, 1:H + 1, 1:W + 1, :] = 0
            assert float(t.abs().max()) == 0.0, "border written"
        print(line, flush=True)
        if H >= 46:
            t_old = gtime(lambda: conv.conv2d(xs, layer))
            line = "   time: old %.1f us" % t_old
            for cfg in (1, 2, 3, 4, 5, 6):
                line += " | cfg%d %.1f" % (cfg, gtime(lambda: c8.conv2d_c8(acts, layer, cfg=cfg)))
            print(line, flush=True)
    # gates
    B, H, W = 1, 184, 312
    h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    x1, x2 = torch.randn(B, 128, H, W, device=DEV), torch.randn(B, 128, H, W, device=DEV)
    cz, cr, cq = (torch.randn(B, 128, H, W, device=DEV) for _ in range(3))
    zr = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
    ql = torch.nn.Conv2d(384, 128, 3, padding=1).to(DEV)
    z0, rh0 = conv.conv2d_gate_zr([h, x1, x2], zr, cz, cr, h)
    hn0 = conv.conv2d_gate_out([rh0, x1, x2], ql, cq, z0, h)
    ah, a1, a2 = c8.pack(h), c8.pack(x1), c8.pack(x2)
    rh_c8 = c8.ActC8(B, 128, H, W, DEV)
    rh = torch.empty_like(h)
    z = c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=rh_c8, rh=rh, cfg=1)
    hn = torch.empty_like(h)
    hn_c8 = c8.ActC8(B, 128, H, W, DEV)
    c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=2)
    print("gates: z %.2e  rh %.2e (c8 %.2e)  h' %.2e (c8 %.2e)" % (rel(z, z0), rel(rh, rh0), rel(c8.unpack(rh_c8), rh0), rel(hn, hn0), rel(c8.unpack(hn_c8), hn0)))
    print("gate timing: old zr %.1f q %.1f | c8 zr %.1f q(cfg2) %.1f q(cfg1) %.1f q(cfg3) %.1f" % (
        gtime(lambda: conv.conv2d_gate_zr([h, x1, x2], zr, cz, cr, h)), gtime(lambda: conv.conv2d_gate_out([rh0, x1, x2], ql, cq, z0, h)),
        gtime(lambda: c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=rh_c8, cfg=1)),
        gtime(lambda: c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=2)),
        gtime(lambda: c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=1)),
        gtime(lambda: c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=3))))
    # tail
    enc = torch.nn.Conv2d(128, 126, 3, padding=1).to(DEV)
    c1, f1 = torch.randn(B, 64, H, W, device=DEV), torch.randn(B, 64, H, W, device=DEV)
    flow = torch.randn(B, 2, H, W, device=DEV)
    mf = c8.ActC8(B, 128, H, W, DEV)
    c8.conv2d_c8([c8.pack(c1), c8.pack(f1)], enc, relu=True, out_c8=mf, tail=flow)
    want = torch.cat([conv.conv2d([c1, f1], enc, relu=True), flow], 1)
    print("tail: %.2e" % rel(c8.unpack(mf), want))


if __name__ == "__main__":
    main()
