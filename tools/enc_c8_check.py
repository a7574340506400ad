"""Encoders with layer1 on conv_c8 (extractor._layer1_c8) vs the round-2 path: output difference, time per encoder call
and per whole pair.  ENC_CFGS="3,4" sweeps the tile shape of the 64 -> 64 layers."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
from dkt_stereo_amd import extractor as ex
DEV = "cuda:0"


def gtime(fn, n=5):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


with torch.no_grad():
    Hh, Ww = (int(v) for v in os.environ.get("ENC_SIZE", "736,1248").split(","))
    m = RAFTStereo()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
    m.to(DEV).eval()
    i1, i2 = _synth.image_pair(3, 1, Hh, Ww, 40)
    i1, i2 = torch.from_numpy(i1).to(DEV), torch.from_numpy(i2).to(DEV)
    x1 = (2 * (i1 / 255.0) - 1.0).contiguous()
    x2 = (2 * (i2 / 255.0) - 1.0).contiguous()
    xx = torch.cat([x1, x2], 0)
    outs = {}
    for c8 in (False, True):
        ex.C8_ENCODER = c8
        f = m.fnet._trunk(xx)
        c = m.cnet._trunk(x1)
        outs[c8] = (f.clone(), c.clone())
        tf = gtime(lambda: m.fnet._trunk(xx))
        tc = gtime(lambda: m.cnet._trunk(x1))
        print("C8_ENCODER=%d: fnet trunk %.3f ms, cnet trunk %.3f ms" % (c8, tf, tc), flush=True)
    for name, k in (("fnet", 0), ("cnet", 1)):
        a, b = outs[True][k].double(), outs[False][k].double()
        print("%s trunk: max|d| %.3e rel-to-max %.3e" % (name, float((a - b).abs().max()), float((a - b).abs().max() / b.abs().max())), flush=True)
    for cfg in [int(v) for v in os.environ.get("ENC_CFGS", "3").split(",")]:
        ex.C8_ENCODER, ex.C8_ENCODER_CFG = True, cfg
        print("cfg %d: fnet %.3f cnet %.3f ms" % (cfg, gtime(lambda: m.fnet._trunk(xx)), gtime(lambda: m.cnet._trunk(x1))), flush=True)
    ex.C8_ENCODER_CFG = 3
    if os.environ.get("ENC_ONLY"):
        sys.exit(0)
    res = {}
    for c8 in (False, True):
        ex.C8_ENCODER = c8
        for _ in range(3):
            _, up = m(i1, i2, iters=32, test_mode=True)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            _, up = m(i1, i2, iters=32, test_mode=True)
        torch.cuda.synchronize()
        res[c8] = (up.clone(), (time.time() - t0) / 5 * 1e3)
    d = (res[True][0] - res[False][0]).abs()
    print("pair: c8-encoder vs round-2 encoder max|d| %.2e ; ms/pair %.2f -> %.2f" % (float(d.max()), res[False][1], res[True][1]), flush=True)
