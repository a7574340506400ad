"""IGEV refinement loop (igev_stereo.py:199-210) below the size from which it used to take loop_c8: ms per 32 iterations on the
round-2 loop (graph + pipelined GRUs) against loop_c8.C8LoopIGEV (two-launch finest GRU on 4-row tiles below 128 tiles, as for
RAFT-Stereo), and the distance between their results.  Decides loop_c8.IGEV_MIN_PIXELS.
    python tools/igev_small_shapes.py"""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
from dkt_stereo_amd import igev_loop, loop_c8  # noqa: E402
from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume  # noqa: E402
from dkt_stereo_amd.update import BasicMultiUpdateBlockIGEV  # noqa: E402

DEV = "cuda:0"
cfg = dict(corr_levels=2, corr_radius=4, n_downsample=2, n_gru_layers=3, hidden_dims=[128, 128, 128], slow_fast_gru=False)
blk = BasicMultiUpdateBlockIGEV(SimpleNamespace(**cfg), hidden_dims=cfg["hidden_dims"])
blk.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(blk), 3))
blk.to(DEV).eval()
iters = 32


def run(H, W, min_pixels):
    loop_c8.IGEV_MIN_PIXELS = min_pixels
    g = torch.Generator(device=DEV).manual_seed(0)
    ml, mr = (torch.randn(1, 96, H, W, device=DEV, generator=g) for _ in range(2))
    net0 = [torch.tanh(torch.randn(1, 128, H >> i, W >> i, device=DEV, generator=g)) for i in range(3)]
    inp = [list((0.5 * torch.randn(1, 384, H >> i, W >> i, device=DEV, generator=g)).split(128, dim=1)) for i in range(3)]
    coords = torch.arange(W, device=DEV).float().view(1, 1, W, 1).repeat(1, H, 1, 1)
    geo = torch.randn(1, 8, 48, H, W, device=DEV, generator=g)
    disp0 = torch.full((1, 1, H, W), 20.0, device=DEV)
    geo_fn = Combined_Geo_Encoding_Volume(ml, mr, geo, radius=4, num_levels=2)
    cache = {}
    on_c8 = loop_c8.eligible_igev(blk, net0[0].shape)
    f = lambda: igev_loop.igev_iterate(blk, geo_fn, disp0, coords, [t.clone() for t in net0], inp, iters, cache=cache)  # noqa: E731
    with torch.no_grad():
        for _ in range(3):
            d, m, _ = f()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            d, m, _ = f()
        torch.cuda.synchronize()
    return (time.time() - t0) / 5 * 1e3, d.clone(), m.clone(), on_c8


for H, W in ((64, 128), (96, 160), (120, 160), (136, 240), (160, 256)):
    t2, d2, m2, c2 = run(H, W, 1 << 30)
    t3, d3, m3, c3 = run(H, W, 0)
    print("1/4-res %3dx%3d (%5d px): round-2 loop %6.2f ms (c8 %s)   loop_c8 %6.2f ms (c8 %s)   max|d disp| %.2e  max|d mask| %.2e" % (
        H, W, H * W, t2, c2, t3, c3, float((d2 - d3).abs().max()), float((m2 - m3).abs().max())), flush=True)
