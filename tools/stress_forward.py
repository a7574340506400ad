"""Determinism stress of the whole forward at the benchmark shape: N forwards of the same 736x1248 pair --
encoders on two streams with their streaming kernels beside MFMA convolutions, the fused ConvGRU hand-off in every
iteration -- must all be bit-identical to the first (torch.equal on the quarter-resolution and the up-sampled disparity).
    python tools/stress_forward.py [N=200] [iters=32]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
from dkt_stereo_amd.raft_stereo import RAFTStereo  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda:0"
model = RAFTStereo()
model.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(model), 7), strict=True)
model.to(dev).eval()
a, b = _synth.image_pair(1000, 1, 736, 1248, 12)
i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
other = [torch.from_numpy(t).to(dev) for t in _synth.image_pair(1001, 1, 736, 1248, 40)]
with torch.no_grad():
    lo0, up0 = model(i1, i2, iters=iters, test_mode=True)
    lo0, up0 = lo0.clone(), up0.clone()
    bad = 0
    t0 = time.time()
    for k in range(N):
        if k % 7 == 3:
            model(other[0], other[1], iters=iters, test_mode=True)        # another pair in between: buffers are reused
        lo, up = model(i1, i2, iters=iters, test_mode=True)
        if not (torch.equal(lo, lo0) and torch.equal(up, up0)):
            bad += 1
            print("forward %d differs: max |d| %.3g" % (k, float((up - up0).abs().max())))
    torch.cuda.synchronize()
print("%d forwards of one 736x1248 pair (%d iterations each): %d differ from the first  [%.1f s]" % (N, iters, bad, time.time() - t0))
print("encoder streams: DKT_ENCODER_STREAMS=%s DKT_CNET_STREAMS=%s; fused ConvGRU: DKT_C8_FUSE_GRU=%s" % (
    os.environ.get("DKT_ENCODER_STREAMS", "1 (default)"), os.environ.get("DKT_CNET_STREAMS", "1 (default)"),
    os.environ.get("DKT_C8_FUSE_GRU", "1 (default)")))
sys.exit(1 if bad else 0)
