"""Fused ConvGRU launch (csrc/gru_c8.hip) against the two-launch form (conv_c8 epilogues 1 + 2) and an fp64 torch
reference; timing of both forms at the cfg2 shapes (gru08 184x312 with gru32 23x39 riding along)."""
import sys
import time

import torch
import torch.nn.functional as F

from dkt_stereo_amd import conv_c8 as c8
from dkt_stereo_amd.update import ConvGRU

DEV = "cuda"


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def make(B, H, W, xch, seed):
    torch.manual_seed(seed)
    gru = ConvGRU(128, sum(xch)).to(DEV)
    h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    xs = [torch.randn(B, c, H, W, device=DEV) for c in xch]
    cz, cr, cq = (torch.randn(B, 128, H, W, device=DEV) for _ in range(3))
    return gru, h, xs, cz, cr, cq


def ref64(gru, h, xs, cz, cr, cq):
    g = ConvGRU(128, sum(x.shape[1] for x in xs)).double().to(DEV)
    g.load_state_dict({k: v.double() for k, v in gru.state_dict().items()})
    h, cz, cr, cq = h.double(), cz.double(), cr.double(), cq.double()
    x = torch.cat([t.double() for t in xs], 1)
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(F.conv2d(hx, g.convz.weight, g.convz.bias, padding=1) + cz)
    r = torch.sigmoid(F.conv2d(hx, g.convr.weight, g.convr.bias, padding=1) + cr)
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), g.convq.weight, g.convq.bias, padding=1) + cq)
    return (1 - z) * h + z * q


class State:
    def __init__(self, gru, h, xs, cz, cr, cq):
        B, _, H, W = h.shape
        self.gru, self.h, self.cz, self.cr, self.cq = gru, h.clone(), cz, cr, cq
        self.hc8 = c8.pack(self.h)
        self.xs = [c8.pack(x) for x in xs]
        self.rh = c8.ActC8(B, 128, H, W, DEV)
        self.flags = c8.gru_flags(B, H, W, DEV)

    def desc(self):
        return c8.gru_desc(self.gru, self.hc8, self.xs, self.rh, self.cz, self.cr, self.cq, self.h, self.flags)

    def two_launch(self, cfg_zr=1, cfg_q=2):
        z = c8.gate_zr([self.hc8, *self.xs], self.gru._merged_zr(), self.cz, self.cr, self.h, rh_c8=self.rh, cfg=cfg_zr)
        c8.gate_out([self.rh, *self.xs], self.gru.convq, self.cq, z, self.h, self.h, out_c8=self.hc8, cfg=cfg_q)


@torch.no_grad()
def check(B, H, W, xch, steps=3, seed=0):
    args = make(B, H, W, xch, seed)
    want = ref64(*args)
    a, b = State(*args), State(*args)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    for i in range(steps):
        ok = c8.gru_launch(a.desc(), err=err)
        if not ok:
            print("  %dx%dx%d: unsupported" % (B, H, W))
            return
        b.two_launch()
        if i == 0:
            print("  B=%d %dx%d x=%s step0: fused vs fp64 %.2e, two-launch vs fp64 %.2e" % (B, H, W, xch, rel(a.h, want), rel(b.h, want)))
        e = (rel(a.h, b.h), rel(c8.unpack(a.hc8), a.h), float((a.hc8.t.float() - b.hc8.t.float()).abs().max()))
        print("    step %d: fused vs two-launch h %.2e, twin vs h %.2e, twin diff %.2e, err word %d" % (i, e[0], e[1], e[2], int(err.item())))
    # determinism: same start twice
    c, d = State(*args), State(*args)
    for i in range(steps):
        c8.gru_launch(c.desc(), err=err)
        c8.gru_launch(d.desc(), err=err)
    print("    deterministic:", torch.equal(c.h, d.h) and torch.equal(c.hc8.t, d.hc8.t), " border zero:",
          float(c.hc8.t[:, :, :, 0].abs().max()) == 0 and float(c.hc8.t[:, :, :, :, 0].abs().max()) == 0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


@torch.no_grad()
def bench():
    big = State(*make(1, 184, 312, [128, 128], 1))
    small = State(*make(1, 23, 39, [128], 2))
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    d0, d1 = big.desc(), small.desc()
    g = torch.cuda.CUDAGraph()

    def graphed(fn):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
        return lambda: g.replay()

    f_single = graphed(lambda: c8.gru_launch(d0, err=err))
    f_pair = graphed(lambda: c8.gru_launch(d0, d1, err=err))
    f_two = graphed(lambda: big.two_launch())
    print("fused gru08 alone      : %.1f us" % (timeit(f_single) / 10))
    print("fused gru08 + gru32    : %.1f us" % (timeit(f_pair) / 10))
    print("two launches gru08     : %.1f us" % (timeit(f_two) / 10))
    print("err word", int(err.item()))


if __name__ == "__main__":
    torch.cuda.init()
    if "bench" not in sys.argv[1:]:
        check(1, 96, 160, [128, 128])
        check(1, 23, 39, [128])
        check(1, 50, 70, [128, 128], seed=3)
        check(2, 184, 312, [128, 128], steps=2, seed=4)      # 460 tiles: two rounds per block
        check(1, 184, 312, [128, 128], steps=4, seed=5)
    if "nobench" not in sys.argv[1:]:
        bench()
