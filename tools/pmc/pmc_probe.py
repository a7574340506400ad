#!/usr/bin/env python3
"""The dispatches whose HBM-side traffic bench.py reports, plus known-traffic calibration kernels, each run a few
times in isolation -- to be wrapped by rocprofv3 --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE (another run):
tools/pmc/run_pmc.sh.  Shapes: BASELINE cfg2 (184x312 at 1/4 resolution), smooth disparities as the GRU loop
produces them."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dkt_stereo_amd import conv  # noqa: E402
from dkt_stereo_amd.corr import CorrBlock1D  # noqa: E402

dev = "cuda:0"
reps = 3
torch.manual_seed(0)
lib = ctypes.CDLL(os.environ.get("PMC_CALIB_LIB", "/tmp/libpmc_calib.so"))
lib.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
with torch.no_grad():
    # ---- calibration: 1 GiB streams (4x the 256 MiB Infinity Cache)
    n = 256 * 1024 * 1024
    src, dst = torch.randn(n, device=dev), torch.empty(n, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for which in range(5):
        for _ in range(reps):
            assert lib.calib_run(which, src.data_ptr(), dst.data_ptr(), n, st) == 0
    torch.cuda.synchronize()
    del src, dst
    # ---- the dominant kernel: gru08 z|r convolution with the gate epilogue
    conv.set_backend("f16x3")
    layer = torch.nn.Conv2d(384, 256, 3, padding=1).to(dev)
    xs = [torch.tanh(torch.randn(1, 128, 184, 312, device=dev))] + [torch.randn(1, 128, 184, 312, device=dev) for _ in range(2)]
    cz, cr = (torch.randn(1, 128, 184, 312, device=dev) for _ in range(2))
    for _ in range(reps):
        conv.conv2d_gate_zr(xs, layer, cz, cr, xs[0])
    # ---- lookup fused with convc1 (B = 1 and B = 8) and the stand-alone lookup
    c1 = torch.nn.Conv2d(36, 64, 1).to(dev)
    for B in (1, 8):
        f1, f2 = (torch.randn(B, 256, 184, 312, device=dev) for _ in range(2))
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        coords = torch.zeros(B, 2, 184, 312, device=dev)
        xs_ = torch.arange(312, device=dev).float().view(1, 1, 312)
        coords[:, 0] = xs_ - (10.0 + 30.0 * xs_ / 312) - 0.3 * torch.rand(B, 184, 312, device=dev)
        for _ in range(reps):
            blk.lookup_conv1x1(coords, c1)
        for _ in range(reps):
            blk(coords)
        del blk, f1, f2
    torch.cuda.synchronize()
print("done")
