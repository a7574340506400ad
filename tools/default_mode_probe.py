#!/usr/bin/env python3
"""Six forwards of the benchmark pair in one mode (argv[1]: "default" = check_finite True, "unchecked") for a rocprofv3 trace."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
from dkt_stereo_amd.raft_stereo import RAFTStereo  # noqa: E402

dev = torch.device("cuda", 0)
m = RAFTStereo()
m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
m.to(dev).eval()
i1, i2 = (torch.from_numpy(t).to(dev) for t in _synth.image_pair(1000, 1, 736, 1248, 12))
with torch.no_grad():
    for _ in range(5):
        m(i1, i2, iters=32, test_mode=True)
    torch.cuda.synchronize()
    m.check_finite = sys.argv[1] == "default"
    t0 = time.perf_counter()
    for _ in range(8):
        m(i1, i2, iters=32, test_mode=True)
    torch.cuda.synchronize()
    print("%s: %.3f ms per pair" % (sys.argv[1], 1e3 * (time.perf_counter() - t0) / 8))
