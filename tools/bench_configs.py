#!/usr/bin/env python3
"""Secondary benchmark lines for BASELINE.json configs[2..4] (bench.py carries configs[1]):

  cfg3  IGEV-Stereo 736x1248: Combined Geometry Encoding volume + 32 GRU iterations from the
        match features / geometry volume onward (the timm feature network and the 3-D
        hourglass cannot be constructed offline) -- ms/iter, pairs/s of the loop
  cfg4  RAFT-Stereo batch 8 per GPU (the per-GPU share of batch 64 over 8 GPUs), 1 GPU
  cfg5  GwcNet 544x960: gwc (40 groups) + concat (2x12) volume, D=48 planes, one fused buffer

    python tools/bench_configs.py [cfg3] [cfg4] [cfg5]
One JSON object per line.
"""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402

DEV = "cuda:0"


def sync_time(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


@torch.no_grad()
def cfg3():
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    from dkt_stereo_amd.submodule import build_gwc_volume
    from dkt_stereo_amd.update import BasicMultiUpdateBlockIGEV
    H, W, iters = 184, 312, 32
    cfg = dict(corr_levels=2, corr_radius=4, n_downsample=2, n_gru_layers=3, hidden_dims=[128, 128, 128],
               slow_fast_gru=False)
    blk = BasicMultiUpdateBlockIGEV(SimpleNamespace(**cfg), hidden_dims=cfg["hidden_dims"])
    blk.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(blk), 3))
    blk.to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(0)
    ml, mr = (torch.randn(1, 96, H, W, device=DEV, generator=g) for _ in range(2))
    net0 = [torch.tanh(torch.randn(1, 128, H >> i, W >> i, device=DEV, generator=g)) for i in range(3)]
    inp = [list((0.5 * torch.randn(1, 384, H >> i, W >> i, device=DEV, generator=g)).split(128, dim=1)) for i in range(3)]
    coords = torch.arange(W, device=DEV).float().view(1, 1, W, 1).repeat(1, H, 1, 1)
    geo = torch.randn(1, 8, 48, H, W, device=DEV, generator=g)   # stands in for the 3-D aggregated volume

    from dkt_stereo_amd.igev_loop import igev_iterate
    cache = {}
    disp0 = torch.full((1, 1, H, W), 20.0, device=DEV)

    def pair():
        build_gwc_volume(ml, mr, 48, 8)                       # igev_stereo.py:169
        if "geo" not in cache:
            cache["geo"] = Combined_Geo_Encoding_Volume(ml, mr, geo, radius=4, num_levels=2)   # :192-193
        else:
            cache["geo"].rebuild(ml, mr, geo)                 # same shapes: pyramids refilled in place
        return igev_iterate(blk, cache["geo"], disp0, coords, net0, inp, iters, cache=cache)[0]   # :199-210

    t = sync_time(pair, 3, 2)
    print(json.dumps({"config": "cfg3 IGEV-Stereo 736x1248 (184x312 @1/4), gwc volume + geometry-encoding "
                                "pyramids + 32 x (geo lookup + IGEV update block); feature/3-D aggregation "
                                "networks excluded (not constructible offline)",
                      "ms_per_pair_loop": 1e3 * t, "ms_per_iter": 1e3 * t / iters, "loop_pairs_per_s": 1.0 / t,
                      "dtype": "f32", "data": "synthetic"}), flush=True)


@torch.no_grad()
def cfg4():
    from dkt_stereo_amd.raft_stereo import RAFTStereo
    B = 8
    m = RAFTStereo()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
    m.to(DEV).eval()
    pairs = [_synth.image_pair(2000 + j, 1, 736, 1248, 12 if j % 2 == 0 else 40) for j in range(B)]
    i1 = torch.cat([torch.from_numpy(p[0]) for p in pairs]).to(DEV)
    i2 = torch.cat([torch.from_numpy(p[1]) for p in pairs]).to(DEV)
    t = sync_time(lambda: m(i1, i2, iters=32, test_mode=True), 2, 2)
    print(json.dumps({"config": "cfg4 RAFT-Stereo 736x1248, 32 iters, batch 8 on one GPU (per-GPU share of "
                                "batch 64 over 8 GPUs)", "ms_per_batch": 1e3 * t, "pairs_per_s": B / t,
                      "dtype": "f32", "data": "synthetic"}), flush=True)


@torch.no_grad()
def cfg5():
    from dkt_stereo_amd.submodule import build_concat_volume, build_gwc_concat_volume, build_gwc_volume
    H, W = 136, 240
    g = torch.Generator(device=DEV).manual_seed(0)
    fl, fr = (torch.randn(1, 320, H, W, device=DEV, generator=g) for _ in range(2))
    cl, cr = (torch.randn(1, 12, H, W, device=DEV, generator=g) for _ in range(2))
    t_sep = sync_time(lambda: torch.cat((build_gwc_volume(fl, fr, 48, 40), build_concat_volume(cl, cr, 48)), 1), 20, 3)
    t_fused = sync_time(lambda: build_gwc_concat_volume(fl, fr, cl, cr, 48, 40), 20, 3)
    out_bytes = 64 * 48 * H * W * 4
    # end to end: dkt_stereo_amd.gwcnet.GWCNet (features on this library's convolutions, fused volume, 3-D
    # aggregation on the vendor library, soft-argmin), 544x960 = 540x960 padded to /32
    from dkt_stereo_amd.gwcnet import GWCNet
    m = GWCNet()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 17))
    m.to(DEV).eval()
    i1, i2 = _synth.image_pair(9, 1, 544, 960, 40)
    i1, i2 = torch.from_numpy(i1).to(DEV), torch.from_numpy(i2).to(DEV)
    torch.backends.cudnn.benchmark = True
    t_e2e = sync_time(lambda: m(i1, i2, test_mode=True), 3, 2)
    feats = {}

    def stage(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        feats[name] = 1e3 * (time.perf_counter() - t0)
        return r
    n1 = (2 * (i1 / 255.0) - 1.0).contiguous()
    n2 = (2 * (i2 / 255.0) - 1.0).contiguous()
    f = stage("features_ms", lambda: m.feature_extraction(torch.cat([n1, n2], 0)))
    fL = {k: v[:1] for k, v in f.items()}
    fR = {k: v[1:] for k, v in f.items()}
    vol = stage("volume_ms", lambda: m.build_volume(fL, fR))
    stage("aggregation_softargmin_ms", lambda: m.cost_regularization(vol))
    print(json.dumps({"config": "cfg5 GwcNet 544x960 (136x240 @1/4): gwc volume 320ch/40 groups + concat volume "
                                "2x12ch, 48 planes -> (1,64,48,136,240); end to end with 3-D aggregation on the vendor library",
                      "us_separate_plus_cat": 1e6 * t_sep, "us_fused_buffer": 1e6 * t_fused,
                      "output_GB_per_s_fused": out_bytes / t_fused / 1e9,
                      "e2e_ms_per_pair": 1e3 * t_e2e, "e2e_pairs_per_s": 1.0 / t_e2e, "e2e_stages": feats,
                      "dtype": "f32", "data": "synthetic"}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    for w in which:
        {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[w]()
