"""Do two kernels of the GRU iteration overlap when they sit on different streams?  (the pair
trace shows the two queues alternating.)  For pairs (A on a side stream, B on the main stream) at cfg2 shapes: time of A
alone, B alone, and of the fork/join {A || B}, eagerly and replayed from a captured graph.  t(A||B) ~ tA + tB means
the pair serialises; ~ max(tA, tB) means it overlaps."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
from dkt_stereo_amd.corr import CorrBlock1D
from dkt_stereo_amd.conv import conv2d
from dkt_stereo_amd.update import interp, pool2x, _leading_outputs
DEV = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def graphed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return lambda: g.replay(), reps


@torch.no_grad()
def main():
    B, H, W = 1, 184, 312
    m = RAFTStereo()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
    m.to(DEV).eval()
    ub = m.update_block
    enc = ub.encoder
    R = lambda *s: torch.randn(*s, device=DEV)
    f1, f2 = R(B, 256, H, W), R(B, 256, H, W)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    xs = torch.arange(W, device=DEV, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H, device=DEV, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
    coords = torch.cat([xs - 20.0 - 3.0 * torch.rand(B, 1, H, W, device=DEV), ys], 1).contiguous()
    flow = R(B, 2, H, W)
    net = [torch.tanh(R(B, 128, H >> i, W >> i)) for i in range(3)]
    inp = [[R(B, 128, H >> i, W >> i) for _ in range(3)] for i in range(3)]
    corr = blk(coords)
    c1 = conv2d(corr, enc.convc1, relu=True)
    fl1 = conv2d(flow, enc.convf1, relu=True)
    p0, u2 = pool2x(net[0]), interp(net[2], net[1])
    y = conv2d(net[0], ub.flow_head.conv1, relu=True)
    big = R(64, 1024, 1024)
    side = torch.cuda.Stream()

    kernels = {
        "gru16": lambda: ub.gru16(net[1], *inp[1], p0, u2),
        "gru08": lambda: ub.gru08(net[0], *inp[0], net[0], net[0]),
        "lookup+convc1": lambda: blk.lookup_conv1x1(coords, enc.convc1),
        "stem7": lambda: conv2d(flow, enc.convf1, relu=True),
        "convc2": lambda: conv2d(c1, enc.convc2, relu=True),
        "enc.conv": lambda: conv2d([c1, fl1], enc.conv, relu=True),
        "fh.conv1": lambda: conv2d(net[0], ub.flow_head.conv1, relu=True),
        "few(x)": lambda: conv2d(y, _leading_outputs(ub.flow_head.conv2, 1)),
        "interp16->8": lambda: interp(net[2], net[1]),
        "interp8->4": lambda: interp(net[1], net[0]),
        "torch mul 256MB": lambda: big.mul_(1.0),
    }
    pairs = [("gru16", "lookup+convc1"), ("gru16", "stem7"), ("gru16", "convc2"), ("gru16", "enc.conv"),
             ("gru16", "few(x)"), ("gru16", "torch mul 256MB"), ("fh.conv1", "interp16->8"),
             ("gru08", "interp16->8"), ("gru08", "lookup+convc1"), ("gru16", "fh.conv1")]
    alone = {}
    for k, fn in kernels.items():
        r, reps = graphed(fn)
        alone[k] = timed(r, 10) / reps
        print("alone  %-18s %8.1f us" % (k, alone[k]), flush=True)

    def forkjoin(a, b):
        def run():
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                kernels[a]()
            kernels[b]()
            main_s.wait_stream(side)
        return run

    for a, b in pairs:
        fj = forkjoin(a, b)
        te = timed(fj, 20)
        r, reps = graphed(fj)
        tg = timed(r, 10) / reps
        print("pair   %-10s || %-18s  sum %7.1f  max %7.1f | eager %7.1f  graph %7.1f us" %
              (a, b, alone[a] + alone[b], max(alone[a], alone[b]), te, tg), flush=True)


if __name__ == "__main__":
    main()
