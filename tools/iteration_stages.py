"""Stand-alone GPU time (HIP-graph replay, so no host launch cost) of every stage of one RAFT-Stereo GRU
iteration at BASELINE cfg2 shapes, and of the whole pipelined iteration, on the real modules."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
from dkt_stereo_amd.corr import CorrBlock1D
from dkt_stereo_amd.conv import conv2d
from dkt_stereo_amd.update import interp, pool2x, harness
DEV = "cuda:0"


def gtime(fn, reps=10, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)


@torch.no_grad()
def main():
    B, H, W = int(os.environ.get("B", 1)), 184, 312
    m = RAFTStereo()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
    m.to(DEV).eval()
    ub = m.update_block
    R = lambda *s: torch.randn(*s, device=DEV)
    f1, f2 = R(B, 256, H, W), R(B, 256, H, W)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    xs = torch.arange(W, device=DEV, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H, device=DEV, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
    coords = torch.cat([xs - 20.0 - 3.0 * torch.rand(B, 1, H, W, device=DEV), ys], 1).contiguous()
    flow = R(B, 2, H, W)
    net = [torch.tanh(R(B, 128, H >> i, W >> i)) for i in range(3)]
    inp = [[R(B, 128, H >> i, W >> i) for _ in range(3)] for i in range(3)]
    enc = ub.encoder
    rows = []
    def T(name, fn):
        us = gtime(fn)
        rows.append((name, us))
        print("%-34s %8.1f us" % (name, us), flush=True)
    corr = blk(coords)
    T("lookup (skew)", lambda: blk(coords))
    T("convc1 1x1 36->64", lambda: conv2d(corr, enc.convc1, relu=True))
    T("lookup+convc1 fused", lambda: blk.lookup_conv1x1(coords, enc.convc1))
    from dkt_stereo_amd.corr import PytorchAlternateCorrBlock1D
    alt = PytorchAlternateCorrBlock1D(f1, f2, num_levels=4, radius=4)
    T("lookup on the fly (alt)", lambda: alt(coords))
    c1 = conv2d(corr, enc.convc1, relu=True)
    T("convc2 64->64", lambda: conv2d(c1, enc.convc2, relu=True))
    T("convf1 7x7 2->64", lambda: conv2d(flow, enc.convf1, relu=True))
    fl1 = conv2d(flow, enc.convf1, relu=True)
    T("convf2 64->64", lambda: conv2d(fl1, enc.convf2, relu=True))
    T("enc.conv 128->126", lambda: conv2d([c1, fl1], enc.conv, relu=True))
    T("motion encoder (all)", lambda: enc(flow, blk.deferred(coords)))
    mf = enc(flow, corr)
    up = interp(net[1], net[0])
    T("interp 1/8->1/4", lambda: interp(net[1], net[0]))
    T("pool2x 1/4->1/8", lambda: pool2x(net[0]))
    T("gru08 (zr+q)", lambda: ub.gru08(net[0], *inp[0], mf, up))
    from dkt_stereo_amd.update import gru_pair
    pl = pool2x(net[1])
    T("gru08 + gru32 paired", lambda: gru_pair(ub.gru08, (net[0], *inp[0], [mf, up], None), ub.gru32, (net[2], *inp[2], [pl], None)))
    T("gru16 (zr+q)", lambda: ub.gru16(net[1], *inp[1], pool2x(net[0]), interp(net[2], net[1])))
    T("gru32 (zr+q)", lambda: ub.gru32(net[2], *inp[2], pool2x(net[1])))
    T("flow_head conv1 128->256", lambda: conv2d(net[0], ub.flow_head.conv1, relu=True))
    y = conv2d(net[0], ub.flow_head.conv1, relu=True)
    T("flow_head conv2 256->2", lambda: conv2d(y, ub.flow_head.conv2))
    from dkt_stereo_amd.update import _leading_outputs
    T("flow_head conv2, x output only", lambda: conv2d(y, _leading_outputs(ub.flow_head.conv2, 1)))
    tot = sum(us for n, us in rows if n not in ("lookup (skew)", "convc1 1x1 36->64", "convc2 64->64", "convf1 7x7 2->64",
                                                 "convf2 64->64", "enc.conv 128->126", "lookup+convc1 fused", "lookup on the fly (alt)", "gru08 + gru32 paired", "flow_head conv2, x output only"))
    print("sum of stand-alone stage times      %8.1f us" % tot)
    # the whole pipelined iteration as the harness runs it
    i1, i2 = _synth.image_pair(3, B, 736, 1248, 40)
    i1, i2 = torch.from_numpy(i1).to(DEV), torch.from_numpy(i2).to(DEV)
    fmap1, fmap2, nl, il = m.encode(i1, i2)
    for _ in range(2):
        m.iterate(fmap1, fmap2, nl, il, 32)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        m.iterate(fmap1, fmap2, nl, il, 32)
    b.record()
    torch.cuda.synchronize()
    print("harness iteration (32-iteration loop / 32) %8.1f us" % (a.elapsed_time(b) * 1e3 / 3 / 32))


if __name__ == "__main__":
    main()
