"""Fused lookup+convc1 at cfg4's per-GPU batch (B = 8) and at B = 1: time and fraction of the 8 TB/s HBM peak (420 B / pixel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c8_check import gtime
from dkt_stereo_amd.corr import CorrBlock1D
from dkt_stereo_amd import conv_c8 as c8
torch.manual_seed(0)
DEV = "cuda:0"
with torch.no_grad():
    for B in (1, 8):
        H, W = 184, 312
        f1, f2 = torch.randn(B, 256, H, W, device=DEV), torch.randn(B, 256, H, W, device=DEV)
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        xs = torch.arange(W, device=DEV, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
        ys = torch.arange(H, device=DEV, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
        coords = torch.cat([xs - 20.0 - 3.0 * torch.rand(B, 1, H, W, device=DEV), ys], 1).contiguous()
        c1 = torch.nn.Conv2d(36, 64, 1).to(DEV)
        dst = c8.ActC8(B, 64, H, W, DEV)
        alg = B * H * W * 420
        t_n = gtime(lambda: blk.lookup_conv1x1(coords, c1), 10, 6)
        t_c = gtime(lambda: blk.lookup_conv1x1(coords, c1, out_c8=dst), 10, 6)
        t_l = gtime(lambda: blk(coords), 10, 6)
        print("B=%d: fused NCHW %.1f us (%.3f of 8 TB/s) | fused C8S %.1f us (%.3f) | stand-alone lookup %.1f us (%.3f on 308 B/px)" % (
            B, t_n, alg / t_n / 8e6, t_c, alg / t_c / 8e6, t_l, B * H * W * 308 / t_l / 8e6), flush=True)
