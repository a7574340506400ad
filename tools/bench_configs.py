#!/usr/bin/env python3
"""Secondary benchmark lines for BASELINE.json configs[2..4] (bench.py carries configs[1]):

  cfg3  IGEV-Stereo 736x1248: Combined Geometry Encoding volume + 32 GRU iterations from the
        match features / geometry volume onward (the timm feature network and the 3-D
        hourglass cannot be constructed offline) -- ms/iter, pairs/s of the loop
  cfg4  RAFT-Stereo batch 8 per GPU (the per-GPU share of batch 64 over 8 GPUs), 1 GPU
  cfg5  GwcNet 544x960: gwc (40 groups) + concat (2x12) volume, D=48 planes, one fused buffer

    python tools/bench_configs.py [cfg3] [cfg4] [cfg5] [--skip-cpu-baseline]
One JSON object per line; cfg3 and cfg5 carry `roofline` (the dominant kernel of the path, timed live) and `cpu_baseline` (the
oracle on this host's cores, bounded sample) as bench.py's line does, and the true `dtype`.
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
    # roofline: the dominant kernel is the fused ConvGRU launch (dkt_gru_c8_pair: gru04 with gru16 of the next iteration riding
    # along), timed where the loop runs it -- the loop's units as plain launches (C8Loop.prologue / unit, what calibrate() runs),
    # an event pair around every fused launch on its stream
    from dkt_stereo_amd import conv_c8 as dc8
    from dkt_stereo_amd.update import harness
    st = cache["state"]
    lp = getattr(st, "c8", None)
    roof = None
    if lp is not None and lp.fuse_gru:
        real, events, used = dc8.gru_launch, [], []

        def timed(d0, d1=None, err=None, ref=None):
            if d1 is None or d0.H != H:
                return real(d0, d1, err=err, ref=ref)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ok = real(d0, d1, err=err, ref=ref)
            b.record()
            if ok:
                events.append((a, b))
                used.append((sum(d0.x_channels[i] for i in range(d0.nx)), d1.H, d1.W, sum(d1.x_channels[i] for i in range(d1.nx))))
            return ok

        d = dict(net=st.net, inp=st.inp, disp=st.disp, coords=st.coords, geo_fn=st.geo_fn)
        dc8.gru_launch = timed
        try:
            with harness(inplace_state=True, side_stream=False):
                st.disp.copy_(disp0)
                for dst, src in zip(st.net, net0):
                    dst.copy_(src)
                lp.prologue(d)
                for k in range(iters):
                    lp.unit(d, last=(k + 1 == iters))
        finally:
            dc8.gru_launch = real
        torch.cuda.synchronize()
        empty = []
        for _ in range(32):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            empty.append((a, b))
        torch.cuda.synchronize()
        over = sorted(x.elapsed_time(y) for x, y in empty)[16]
        ms = [max(a.elapsed_time(b) - over, 1e-6) for a, b in events]
        xc, h2, w2, xc2 = used[0]
        flops = 2.0 * H * W * (128 + xc) * 9 * 384 + 2.0 * h2 * w2 * (128 + xc2) * 9 * 384
        avg = sum(ms) / len(ms)
        ach = flops / (avg * 1e-3) / 1e12
        roof = {"kernel": "gru_c8_kernel (dkt_gru_c8_pair): gru04 z|r %d->256 + gates + q %d->128 + state update @%dx%d, gru16 "
                          "(%d->256, %d->128 @%dx%d) riding along" % (128 + xc, 128 + xc, H, W, 128 + xc2, 128 + xc2, h2, w2),
                "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "mfma_passes": 3,
                "mfma_issue_frac": 3 * ach / 2500.0, "algorithmic_flops_per_launch": flops, "avg_launch_us": 1e3 * avg,
                "launches_timed": len(ms), "traffic": None,
                "traffic_note": "not measured for this loop (RAFT's launch of the same kernel: profiles/r05_hbm_traffic.txt)"}
    # CPU baseline: the oracle's restatement of igev_stereo.py:192-210 (pinned by tests/golden/igev_loop.npz) on this host's
    # cores, bounded sample: pyramids once, `n_cpu` iterations, extrapolated to 32
    cpu = None
    if "--skip-cpu-baseline" not in sys.argv:
        from oracle import torch_oracle as to
        sd = {"update_block." + k: v.cpu() for k, v in blk.state_dict().items()}
        n_cpu = 4
        args_cpu = (sd, dict(cfg), ml.cpu(), mr.cpu(), geo.cpu(), disp0.cpu(), [t.cpu() for t in net0], [[t.cpu() for t in s_] for s_ in inp])
        to.igev_iterations(*args_cpu, 1)
        t0 = time.perf_counter()
        to.igev_iterations(*args_cpu, 1)
        t1 = time.perf_counter()
        to.igev_iterations(*args_cpu, n_cpu)
        t2 = time.perf_counter()
        per_iter = max((t2 - t1) - (t1 - t0), 1e-9) / (n_cpu - 1)
        pair_s = (t1 - t0) + (iters - 1) * per_iter
        cpu = {"value": 1.0 / pair_s, "unit": "pairs/s (loop)", "cores": torch.get_num_threads(), "kind": "port",
               "sample": "geometry pyramids once + %d of %d iterations timed (%.2f s/iter), extrapolated" % (n_cpu, iters, per_iter),
               "s_per_pair": pair_s}
    print(json.dumps({"config": "cfg3 IGEV-Stereo 736x1248 (184x312 @1/4), gwc volume + geometry-encoding "
                                "pyramids + 32 x (geo lookup + IGEV update block); feature/3-D aggregation "
                                "networks excluded (not constructible offline)",
                      "metric": "stereo pairs/sec (refinement loop from the match features / geometry volume on)",
                      "value": 1.0 / t, "unit": "pairs/s", "higher_is_better": True,
                      "ms_per_pair_loop": 1e3 * t, "ms_per_iter": 1e3 * t / iters, "loop_pairs_per_s": 1.0 / t,
                      "dtype": "f32 io/accumulate; convolution products = 3x fp16-split MFMA (22-bit operands); geometry lookup, "
                               "gwc volume, correlation exact fp32",
                      "roofline": roof, "cpu_baseline": cpu, "data": "synthetic"}), flush=True)


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
                      "dtype": "f32 io/accumulate; convolution products = 3x fp16-split MFMA (22-bit operands); lookup / correlation "
                               "exact fp32 (roofline and cpu_baseline: `bench.py --batch 8`)", "data": "synthetic"}), flush=True)


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
    # roofline: the path's own kernel here is the group-wise correlation on the matrix pipe (dkt_gwc_volume_mfma inside the fused
    # gwc + concat buffer): HBM-bound by its plane stores.  Algorithmic bytes = both feature maps read once + every plane written
    # once; launch time from a graph of back-to-back launches (the kernel is stand-alone in the pipeline too)
    vol_g = torch.empty(1, 40, 48, H, W, device=DEV)
    from dkt_stereo_amd import submodule as sm

    def one():
        sm.build_gwc_volume(fl, fr, 48, 40)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    gl = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gl):
        for _ in range(10):
            one()
    gl.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gl.replay()
    e1.record()
    torch.cuda.synchronize()
    gwc_ms = e0.elapsed_time(e1) / 100.0
    alg = 2 * 320 * H * W * 4 + 40 * 48 * H * W * 4
    del vol_g
    roof = {"kernel": "gwc_mfma_kernel (dkt_gwc_volume_mfma): 320 channels / 40 groups, 48 planes @%dx%d, exact-fp32 MFMA" % (H, W),
            "bound": "hbm", "achieved": alg / (gwc_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": alg / (gwc_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": alg, "avg_launch_us": 1e3 * gwc_ms,
            "launches_timed": 100, "traffic": None, "traffic_note": "PMC passes of this kernel: profiles/r03_gwc_mfma.txt"}
    cpu = None
    if "--skip-cpu-baseline" not in sys.argv:
        from oracle import torch_oracle as to
        a_ = (fl.cpu(), fr.cpu(), cl.cpu(), cr.cpu())
        to.gwc_volume(a_[0], a_[1], 48, 40)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.cat((to.gwc_volume(a_[0], a_[1], 48, 40), to.concat_volume(a_[2], a_[3], 48, True)), 1)
        tc = (time.perf_counter() - t0) / 3
        cpu = {"value": 1.0 / tc, "unit": "volumes/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": "gwc (40 groups) + concat volume + cat, 3 repetitions, %.3f s each" % tc, "us_per_volume": 1e6 * tc}
    print(json.dumps({"config": "cfg5 GwcNet 544x960 (136x240 @1/4): gwc volume 320ch/40 groups + concat volume "
                                "2x12ch, 48 planes -> (1,64,48,136,240); end to end with 3-D aggregation on the vendor library",
                      "metric": "cost volumes/sec (gwc + concat into one buffer)", "value": 1.0 / t_fused, "unit": "volumes/s",
                      "higher_is_better": True,
                      "us_separate_plus_cat": 1e6 * t_sep, "us_fused_buffer": 1e6 * t_fused,
                      "output_GB_per_s_fused": out_bytes / t_fused / 1e9,
                      "e2e_ms_per_pair": 1e3 * t_e2e, "e2e_pairs_per_s": 1.0 / t_e2e, "e2e_stages": feats,
                      "dtype": "f32: volumes exact fp32 (group-wise correlation on the fp32 matrix pipe); feature network: "
                               "convolution products = 3x fp16-split MFMA (22-bit operands); 3-D aggregation: vendor fp32",
                      "roofline": roof, "cpu_baseline": cpu, "data": "synthetic"}), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg3", "cfg4", "cfg5"]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    if len(which) > 1:
        # one process per configuration: the CPU baselines leave 100+ OpenMP threads behind, and the next configuration's
        # launches would be timed beside them (round 6: cfg4 measured 562 ms per batch behind cfg3's CPU sample, 236 alone)
        import subprocess
        for w in which:
            subprocess.run([sys.executable, os.path.abspath(__file__), w] + flags, check=False)
    else:
        {"cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[which[0]]()
