#!/usr/bin/env python3
"""Round 5: the dispatches whose HBM-side traffic bench.py reports, taken from THE LOOP ITSELF (round 4's
probe launched the fused ConvGRU step with a 23x39 rider where the loop's is 46x78): one 736x1248 / 32-iteration forward of the
benchmark model with its units as plain launches (model.c8_eager), after the forwards that calibrate and capture -- the LAST 32
gru_c8_kernel and 31 motion_front_kernel dispatches of the process are the loop's own -- plus the reference-visible lookup
operator and the known-traffic calibration kernels; wrapped by rocprofv3 --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE
(another): tools/pmc/run_pmc.sh [batch]."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
from dkt_stereo_amd.raft_stereo import RAFTStereo  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = ctypes.CDLL(os.environ.get("PMC_CALIB_LIB", "/tmp/libpmc_calib.so"))
lib.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
with torch.no_grad():
    n = 256 * 1024 * 1024
    src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for which in range(5):
        for _ in range(3):
            assert lib.calib_run(which, src.data_ptr(), dst.data_ptr(), n, st) == 0
    torch.cuda.synchronize()
    del src, dst
    model = RAFTStereo()
    model.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(model), 7), strict=True)
    model.to(dev).eval()
    pairs = [_synth.image_pair(1000 + j, 1, 736, 1248, 12 if j % 2 == 0 else 40) for j in range(B)]
    i1 = torch.cat([torch.from_numpy(p[0]) for p in pairs]).to(dev)
    i2 = torch.cat([torch.from_numpy(p[1]) for p in pairs]).to(dev)
    model(i1, i2, iters=32, test_mode=True)          # calibrates, packs, captures
    model.c8_eager = True
    model(i1, i2, iters=32, test_mode=True)          # the probed forward: every unit as plain launches
    model.c8_eager = False
    lp = model._graph_state["c8"]
    assert lp.fuse_gru and lp.front and not lp.take_error()
    gs = model._graph_state
    for _ in range(3):
        gs["corr"](gs["coords1"])                     # corr1d_lookup_skew_kernel on the final coordinates
    torch.cuda.synchronize()
print("done rider %dx%d" % tuple(gs["net"][2].shape[2:]))
