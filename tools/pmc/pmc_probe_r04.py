#!/usr/bin/env python3
"""Round 4: the dispatches whose HBM-side traffic bench.py reports -- the fused ConvGRU launch (gru_c8.hip: gru08 at 184x312
with gru32 at 23x39 riding along, exactly the loop's launch) and the fused lookup + convc1 writing C8S (corr_feat64_kernel,
B = 1 and 8) -- plus the known-traffic calibration kernels, each a few times in isolation; wrapped by rocprofv3 --pmc
FETCH_SIZE (one run) and --pmc WRITE_SIZE (another): tools/pmc/run_pmc_r04.sh."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dkt_stereo_amd import conv, conv_c8 as c8  # noqa: E402
from dkt_stereo_amd.corr import CorrBlock1D  # noqa: E402
import gru_c8_check as chk  # noqa: E402

dev = "cuda:0"
reps = 3
torch.manual_seed(0)
lib = ctypes.CDLL(os.environ.get("PMC_CALIB_LIB", "/tmp/libpmc_calib.so"))
lib.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
with torch.no_grad():
    n = 256 * 1024 * 1024
    src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for which in range(5):
        for _ in range(reps):
            assert lib.calib_run(which, src.data_ptr(), dst.data_ptr(), n, st) == 0
    torch.cuda.synchronize()
    del src, dst
    conv.set_backend("f16x3")
    H, W = 184, 312
    big = chk.State(*chk.make(1, H, W, [128, 128], 1))
    small = chk.State(*chk.make(1, 23, 39, [128], 2))
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    for _ in range(reps):
        assert c8.gru_launch(big.desc(), small.desc(), err=err)
    c1 = torch.nn.Conv2d(36, 64, 1).to(dev)
    for B in (1, 8):
        f1, f2 = (torch.randn(B, 256, H, W, device=dev) for _ in range(2))
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        coords = torch.zeros(B, 2, H, W, device=dev)
        xs_ = torch.arange(W, device=dev).float().view(1, 1, W)
        coords[:, 0] = xs_ - (10.0 + 30.0 * xs_ / W) - 0.3 * torch.rand(B, H, W, device=dev)
        dst = c8.ActC8(B, 64, H, W, dev)
        for _ in range(reps):
            assert blk.lookup_conv1x1(coords, c1, out_c8=dst) is not None
        del blk, f1, f2
    torch.cuda.synchronize()
print("done")
