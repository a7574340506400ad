#!/usr/bin/env python3
"""Per-kernel microbenchmarks on one MI355X (run through gpurun).  Every number
is HIP-event time over N back-to-back launches on torch's current stream
(the stream handed to the C ABI), reported per launch together with the
algorithmic bytes / flops of SURVEY.md section 8d.

    python tools/bench_kernels.py [lookup] [build] [gates] [conv] [e2e] [autocast] [volumes]
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402

DEV = "cuda:0"
RESULTS = {}


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def timeit(fn, n=50, warm=5, graph=True):
    """Microseconds per call.  The calls are captured into a HIP graph (10 per replay) so that host
    launch cost (15-20 us per Python-level call) does not hide kernels shorter than that; callables
    that cannot be captured (autograd, host syncs) are timed as back-to-back eager launches."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph and os.environ.get("DKT_BENCH_EAGER", "0") != "1":
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            reps = max(1, n // 10)
            a.record()
            for _ in range(reps):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / (reps * 10) * 1e3
        except Exception:  # noqa: BLE001
            torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3     # microseconds


def report(name, us, bytes_=None, flops=None):
    line = "%-44s %10.2f us" % (name, us)
    r = {"us": us}
    if bytes_:
        r["GBps"] = bytes_ / us / 1e3
        line += "  %8.1f GB/s (%.1f%% of 8 TB/s)" % (r["GBps"], r["GBps"] / 80.0)
    if flops:
        r["TFLOPs"] = flops / us / 1e6
        line += "  %8.1f TFLOP/s" % r["TFLOPs"]
    RESULTS[name] = r
    print(line, flush=True)


@torch.no_grad()
def bench_lookup():
    from dkt_stereo_amd.corr import CorrBlock1D
    for B in (1, 8):
        H, W, C = 184, 312, 256
        f1, f2 = (torch.randn(B, C, H, W, device=DEV) for _ in range(2))
        blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
        n = B * H * W
        alg = n * 308
        for kind in ("smooth", "random"):
            coords = torch.zeros(B, 2, H, W, device=DEV)
            xs = torch.arange(W, device=DEV).float().view(1, 1, W)
            if kind == "smooth":
                coords[:, 0] = xs - 20.3 - 3.0 * torch.sin(torch.arange(H, device=DEV).float() / 9.0).view(1, H, 1)
            else:
                coords[:, 0] = xs - 60.0 * torch.rand(B, H, W, device=DEV)
            from dkt_stereo_amd.corr import _lookup
            for variant in ("1", "4"):
                os.environ["DKT_LOOKUP_VARIANT"] = variant
                report("lookup rows v%s B=%d %s" % (variant, B, kind),
                       timeit(lambda: _lookup(blk.corr_pyramid, coords, 4, W), n=200), bytes_=alg)
            report("lookup skew    B=%d %s" % (B, kind), timeit(lambda: blk(coords), n=200), bytes_=alg)
            c1 = torch.nn.Conv2d(36, 64, 1).to(DEV)
            report("lookup + convc1 fused B=%d %s" % (B, kind), timeit(lambda: blk.lookup_conv1x1(coords, c1), n=200),
                   bytes_=n * 420)
            if kind == "smooth":
                from dkt_stereo_amd.corr import PytorchAlternateCorrBlock1D
                coords[:, 1] = torch.arange(H, device=DEV).float().view(1, H, 1)
                alt = PytorchAlternateCorrBlock1D(f1, f2, num_levels=4, radius=4)
                # algorithmic bytes: both feature maps once per level pyramid (f1 x L, pooled f2 once) + coords + output
                alt_bytes = B * C * H * W * 4 * (4 + 1.875) + n * (8 + 144)
                report("lookup on the fly (alt) B=%d %s" % (B, kind), timeit(lambda: alt(coords), n=50), bytes_=alt_bytes)
                del alt
        os.environ.pop("DKT_LOOKUP_VARIANT", None)
        del blk, f1, f2


@torch.no_grad()
def bench_build():
    from dkt_stereo_amd.corr import CorrBlock1D
    B, C, H, W = 1, 256, 184, 312
    f1, f2 = (torch.randn(B, C, H, W, device=DEV) for _ in range(2))
    fl = 2.0 * B * H * W * W * C
    by = 2 * B * C * H * W * 4 + B * H * W * 4 * sum(W >> i for i in range(4))
    report("corr1d_build + corr1d_skew cfg2 (CorrBlock1D ctor)", timeit(lambda: CorrBlock1D(f1, f2, num_levels=4, radius=4), n=20), bytes_=by, flops=fl)
    report("torch einsum+pools (rocBLAS) cfg2",
           timeit(lambda: _torch_build(f1, f2), n=10), flops=fl)


def _torch_build(f1, f2):
    b, c, h, w = f1.shape
    vol = torch.einsum('aijk,aijh->ajkh', f1, f2).reshape(b * h * w, 1, 1, w) / 16.0
    out = [vol]
    for _ in range(3):
        vol = F.avg_pool2d(vol, [1, 2], stride=[1, 2])
        out.append(vol)
    return out


@torch.no_grad()
def bench_gates():
    from dkt_stereo_amd import _ffi
    L = _ffi.lib()
    B, Ch, H, W = 1, 128, 184, 312
    HW = H * W
    azr = torch.randn(B, 2 * Ch, H, W, device=DEV)
    ctx = torch.randn(B, 3 * Ch, H, W, device=DEV)
    cz, cr, cq = ctx.split(Ch, dim=1)
    h = torch.tanh(torch.randn(B, Ch, H, W, device=DEV))
    z = torch.empty_like(h)
    rh = torch.empty_like(h)
    aq = torch.randn_like(h)
    out = torch.empty_like(h)
    st = _ffi.stream_of(h)
    plane = B * Ch * HW * 4
    report("gru_gate_zr 184x312", timeit(lambda: L.dkt_gru_gate_zr(
        azr.data_ptr(), cz.data_ptr(), cz.stride(0), cr.data_ptr(), cr.stride(0), h.data_ptr(), h.stride(0),
        z.data_ptr(), rh.data_ptr(), rh.stride(0), B, Ch, HW, 0, st), n=100), bytes_=7 * plane)
    report("gru_gate_out 184x312", timeit(lambda: L.dkt_gru_gate_out(
        aq.data_ptr(), cq.data_ptr(), cq.stride(0), z.data_ptr(), h.data_ptr(), h.stride(0),
        out.data_ptr(), out.stride(0), B, Ch, HW, 0, st), n=100), bytes_=5 * plane)


CONV_LAYERS = [
    # name, src channels, Cout, H, W, k, relu
    ("gru08.zr 384->256 @184x312", [128, 128, 128], 256, 184, 312, 3, False),
    ("gru08.q  384->128 @184x312", [128, 128, 128], 128, 184, 312, 3, False),
    ("gru16.zr 384->256 @92x156", [128, 128, 128], 256, 92, 156, 3, False),
    ("gru32.zr 256->256 @46x78", [128, 128], 256, 46, 78, 3, False),
    ("enc.convc1 36->64 1x1", [36], 64, 184, 312, 1, True),
    ("enc.convc2 64->64", [64], 64, 184, 312, 3, True),
    ("enc.conv 128->126", [64, 64], 126, 184, 312, 3, True),
    ("flow_head.conv1 128->256", [128], 256, 184, 312, 3, True),
    ("flow_head.conv2 256->2", [256], 2, 184, 312, 3, False),
    ("mask.2 256->144 1x1", [256], 144, 184, 312, 1, False),
]


@torch.no_grad()
def bench_conv():
    from dkt_stereo_amd import conv
    torch.backends.cudnn.benchmark = True
    for name, chans, cout, H, W, k, relu in CONV_LAYERS:
        cin = sum(chans)
        layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(DEV)
        xs = [torch.randn(1, c, H, W, device=DEV) for c in chans]
        xcat = torch.cat(xs, 1)
        fl = 2.0 * H * W * cin * k * k * cout
        ref = F.conv2d(xcat.double(), layer.weight.double(), layer.bias.double(), padding=k // 2)
        if relu:
            ref = ref.clamp_min(0)
        for be in ("miopen", "f16x3", "f16x2", "f16"):
            conv.set_backend(be)
            arg = xcat if be == "miopen" else (xs if len(xs) > 1 else xs[0])
            us = timeit(lambda: conv.conv2d(arg, layer, relu=relu), n=20, warm=3)
            err = float((conv.conv2d(arg, layer, relu=relu).double() - ref).abs().max() / ref.abs().max())
            report("conv %-30s %-6s" % (name, be), us, flops=fl)
            RESULTS["conv %-30s %-6s" % (name, be)]["rel_err"] = err
            print("     rel err %.2e" % err)
        conv.set_backend("miopen")


@torch.no_grad()
def bench_ablate():
    """Timing-only ablations of the flagship conv shape (gru08 z|r): which part costs what."""
    from dkt_stereo_amd import conv
    conv.set_backend("f16x3")
    layer = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
    xs = [torch.randn(1, 128, 184, 312, device=DEV) for _ in range(3)]
    names = {0: "full", 1: "no weight loads", 2: "no LDS fragment reads", 4: "no staging",
             8: "no MFMA", 7: "MFMA only (no loads/staging)"}
    for abl, nm in names.items():
        os.environ["DKT_CONV_ABLATE"] = str(abl)
        report("ablate zr conv: %s" % nm, timeit(lambda: conv.conv2d(xs, layer), n=20, warm=3))
    os.environ.pop("DKT_CONV_ABLATE", None)
    conv.set_backend("miopen")


def _model(backend="miopen"):
    from dkt_stereo_amd import conv
    from dkt_stereo_amd.raft_stereo import RAFTStereo
    conv.set_backend(backend)
    m = RAFTStereo()
    sd = _synth.torch_state_dict(_synth.shapes_of(m), 7)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


@torch.no_grad()
def bench_e2e():
    from dkt_stereo_amd import conv
    torch.backends.cudnn.benchmark = True
    i1, i2 = _synth.image_pair(1000, 1, 736, 1248, 12)
    i1, i2 = G(i1), G(i2)
    base = None
    for be in ("miopen", "f16x3", "f16x2", "f16"):
        m = _model(be)
        for _ in range(2):
            _, up = m(i1, i2, iters=32, test_mode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _, up = m(i1, i2, iters=32, test_mode=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        fm = m.encode(i1, i2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.iterate(*fm, 32)
        torch.cuda.synchronize()
        hot = (time.perf_counter() - t0) * 1e3
        if base is None:
            base = up.clone()
        d = float((up - base).abs().max())
        epe = float((up - base).abs().mean())
        print("e2e 736x1248 32 iters  %-7s %8.2f ms/pair (%.2f pairs/s)  hot path %.2f ms  "
              "max|d vs miopen| %.3e  mean %.3e" % (be, ms, 1e3 / ms, hot, d, epe), flush=True)
        RESULTS["e2e " + be] = {"ms_per_pair": ms, "hot_ms": hot, "max_abs_vs_miopen": d, "mean_abs_vs_miopen": epe}
    conv.set_backend("miopen")


@torch.no_grad()
def bench_autocast():
    """Vendor convolutions under torch autocast (what the reference's mixed_precision
    flag does, raft_stereo.py:156): speed and drift of the update block in half precision."""
    i1, i2 = _synth.image_pair(1000, 1, 736, 1248, 12)
    i1, i2 = G(i1), G(i2)
    m = _model("miopen")
    fm = m.encode(i1, i2)
    _, base = m.iterate(*fm, 32)
    for dt in (torch.float16, torch.bfloat16):
        try:
            with torch.autocast("cuda", dtype=dt):
                for _ in range(2):
                    _, up = m.iterate(*fm, 32)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, up = m.iterate(*fm, 32)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
            print("autocast %s: hot path %.2f ms, max|d| %.3e mean %.3e" % (
                dt, ms, float((up.float() - base).abs().max()), float((up.float() - base).abs().mean())), flush=True)
        except Exception as e:  # noqa: BLE001
            print("autocast %s failed: %s" % (dt, str(e)[:200]))


@torch.no_grad()
def bench_volumes():
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    from dkt_stereo_amd.submodule import build_concat_volume, build_gwc_volume
    a, b = (torch.randn(1, 96, 184, 312, device=DEV) for _ in range(2))
    report("gwc_volume IGEV 96ch G=8 D=48", timeit(lambda: build_gwc_volume(a, b, 48, 8), n=20),
           bytes_=2 * a.numel() * 4 + 8 * 48 * 184 * 312 * 4)
    a, b = (torch.randn(1, 320, 136, 240, device=DEV) for _ in range(2))
    report("gwc_volume GwcNet 320ch G=40 D=48", timeit(lambda: build_gwc_volume(a, b, 48, 40), n=20),
           bytes_=2 * a.numel() * 4 + 40 * 48 * 136 * 240 * 4)
    a, b = (torch.randn(1, 12, 136, 240, device=DEV) for _ in range(2))
    report("concat_volume GwcNet 12ch D=48", timeit(lambda: build_concat_volume(a, b, 48), n=20),
           bytes_=2 * a.numel() * 4 + 24 * 48 * 136 * 240 * 4)
    m1, m2 = (torch.randn(1, 96, 184, 312, device=DEV) for _ in range(2))
    geo = torch.randn(1, 8, 48, 184, 312, device=DEV)
    fn = Combined_Geo_Encoding_Volume(m1, m2, geo, num_levels=2, radius=4)
    disp = torch.rand(1, 1, 184, 312, device=DEV) * 40
    coords = torch.arange(312, device=DEV).float().view(1, 1, 312, 1).repeat(1, 184, 1, 1)
    report("geo_lookup IGEV cfg3 (random per-pixel disparity)", timeit(lambda: fn(disp, coords), n=50), bytes_=184 * 312 * 1376)
    xs = torch.arange(312, device=DEV).float().view(1, 1, 1, 312)
    smooth = (8.0 + 24.0 * xs / 312 + 0.3 * torch.rand(1, 1, 184, 312, device=DEV)).contiguous()
    report("geo_lookup IGEV cfg3 (smooth disparity)", timeit(lambda: fn(smooth, coords), n=50), bytes_=184 * 312 * 1376)
    # fused with the motion encoder's 1x1 layer (162 -> 64): algorithmic bytes = the pyramids' windows + disp + coords + output
    from dkt_stereo_amd import conv_c8 as c8
    c1 = torch.nn.Conv2d(162, 64, 1).to(DEV)
    alg = 184 * 312 * (162 * 2 * 4 * 10 // 9 // 2 + 8 + 256)
    dst = c8.ActC8(1, 64, 184, 312, DEV)
    report("geo_lookup + convc1 fused -> C8S (smooth)", timeit(lambda: fn.lookup_conv1x1(smooth, coords, c1, out_c8=dst), n=50), bytes_=184 * 312 * (728 + 256))
    report("geo_lookup + convc1 fused -> fp32 (smooth)", timeit(lambda: fn.lookup_conv1x1(smooth, coords, c1), n=50), bytes_=184 * 312 * (728 + 256))
    report("geo_lookup + convc1 fused -> C8S (random)", timeit(lambda: fn.lookup_conv1x1(disp, coords, c1, out_c8=dst), n=50), bytes_=184 * 312 * (728 + 256))


def bench_next():
    """SURVEY 8f rows at cfg2 / cfg3 sizes: PCVNet lookup, CGI normalised volume, lookup / pyramid
    backward, fused convex up-sampling (vs the torch op sequence)."""
    import torch.nn.functional as F
    from dkt_stereo_amd.corr import CorrBlock1D
    from dkt_stereo_amd.pcvnet_corr import CorrBlock1D as PcvBlock
    from dkt_stereo_amd.raft_stereo import RAFTStereo, make_args
    from dkt_stereo_amd.submodule import build_norm_correlation_volume
    H, W = 184, 312
    g = torch.Generator(device=DEV).manual_seed(0)
    with torch.no_grad():
        f1, f2 = (torch.randn(1, 256, H, W, device=DEV, generator=g) for _ in range(2))
        Gn, S, L = 2, 9, 3
        blk = PcvBlock(f1, f2, sample_num=S, num_levels=L, downsample=2)
        base = torch.arange(W, device=DEV).float().view(1, 1, 1, W).expand(1, Gn, H, W)
        coords = (base - 40 * torch.rand(1, Gn, H, W, device=DEV, generator=g)).contiguous()
        sigma = (0.5 + 2 * torch.rand(1, Gn, H, W, device=DEV, generator=g)).contiguous()
        alg = H * W * (L * Gn * S * (8 + 4) + Gn * 8)       # 2 floats read + 1 written per tap, coords + sigma
        report("pcv_lookup G=2 S=9 L=3 (factor 4)", timeit(lambda: blk(coords, sigma), n=100), bytes_=alg)
        a, b = (torch.randn(1, 96, H, W, device=DEV, generator=g) for _ in range(2))
        alg = 2 * 2 * 96 * H * W * 4 + 2 * 96 * H * W * 4 + 48 * H * W * 4   # normalise (r+w) both maps, read them, write volume
        report("norm_correlation_volume 96ch D=48", timeit(lambda: build_norm_correlation_volume(a, b, 48), n=50), bytes_=alg)
        flow = torch.randn(1, 2, H, W, device=DEV, generator=g)
        mask = torch.randn(1, 144, H, W, device=DEV, generator=g)
        fake = type("M", (), {"args": make_args(n_downsample=2)})()
        alg = (144 + 2) * H * W * 4 + 2 * 16 * H * W * 4
        report("convex_upsample fused (dkt)", timeit(lambda: RAFTStereo.upsample_flow(fake, flow, mask), n=100), bytes_=alg)

        def torch_up():
            m = torch.softmax(mask.view(1, 1, 9, 4, 4, H, W), dim=2)
            up = F.unfold(4 * flow, [3, 3], padding=1).view(1, 2, 9, 1, 1, H, W)
            return torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(1, 2, 4 * H, 4 * W)
        report("convex_upsample torch op sequence", timeit(torch_up, n=50), bytes_=alg)
    # backward: 1 lookup's gradient -> level grads (zero-fill + scatter), then pooled chain + 2 GEMMs
    a1 = f1.clone().requires_grad_(True)
    b1 = f2.clone().requires_grad_(True)
    cb = CorrBlock1D(a1, b1, num_levels=4, radius=4)
    xy = torch.zeros(1, 2, H, W, device=DEV)
    xy[:, 0] = torch.arange(W, device=DEV).float().view(1, 1, W) - 30 * torch.rand(1, H, W, device=DEV, generator=g)
    out = cb(xy)
    go = torch.randn_like(out)
    pyr_bytes = 4 * H * W * (312 + 156 + 78 + 39)
    report("lookup backward (zero-fill + scatter, 4 levels)",
           timeit(lambda: torch.autograd.grad(out, cb.corr_pyramid, go, retain_graph=True), n=30, graph=False),
           bytes_=pyr_bytes + out.numel() * 4)
    report("lookup + pyramid + corr backward -> grad fmaps",
           timeit(lambda: torch.autograd.grad(out, [a1, b1], go, retain_graph=True), n=20, graph=False),
           flops=2 * 2.0 * H * W * W * 256)


@torch.no_grad()
def bench_c8():
    """Round 3: the C8S convolution (conv_c8.hip) per layer class of the refinement loop at cfg2 sizes, its producers, the
    lookup writing C8S, the MFMA group-wise correlation, and the Winograd go/no-go timing proxy."""
    from dkt_stereo_amd import conv_c8 as c8
    from dkt_stereo_amd import submodule as sm
    from dkt_stereo_amd.corr import CorrBlock1D
    H, W = 184, 312
    torch.manual_seed(0)
    mk = lambda cin, cout: torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)

    def acts(n, h, w, c=128):
        return [c8.pack(torch.randn(1, c, h, w, device=DEV)) for _ in range(n)]

    fl = lambda cin, cout, h, w: 2.0 * 9 * cin * cout * h * w * 3          # MFMA flops: three split passes
    # gru08 z|r (tile <4,2,4,4>) and q (<2,4,2,4>) with their gate epilogues
    a3 = acts(3, H, W)
    hh = torch.tanh(torch.randn(1, 128, H, W, device=DEV))
    cz, cr, cq = (torch.randn(1, 128, H, W, device=DEV) for _ in range(3))
    rh, hc = c8.ActC8(1, 128, H, W, DEV), c8.ActC8(1, 128, H, W, DEV)
    zr, q = mk(384, 256), mk(384, 128)
    z = c8.gate_zr(a3, zr, cz, cr, hh, rh_c8=rh, cfg=1)
    report("conv_c8 gru08 z|r 384->256 + gates (cfg1)", timeit(lambda: c8.gate_zr(a3, zr, cz, cr, hh, rh_c8=rh, cfg=1), n=100), flops=fl(384, 256, H, W))
    hn = torch.empty_like(hh)
    report("conv_c8 gru08 q 384->128 + update (cfg2)", timeit(lambda: c8.gate_out(a3, q, cq, z, hh, hn, out_c8=hc, cfg=2), n=100), flops=fl(384, 128, H, W))
    # Winograd F(2x2,3x3) timing proxy: the same kernel with 16/36 of the K steps (Cin 176 of 384) has the MFMA count of a
    # Winograd form of the z|r layer with ideal operand reuse -- an upper bound on what the transform could buy
    a176 = [c8.pack(torch.randn(1, 176, H, W, device=DEV))]
    zr176 = mk(176, 256)
    report("  proxy: same layer with 16/36 of the MFMAs (Cin 176)", timeit(lambda: c8.gate_zr(a176, zr176, cz, cr, hh, rh_c8=rh, cfg=1), n=100), flops=fl(176, 256, H, W))
    # gru16 (92x156), gru32 (46x78): 256 / 384 input channels
    for name, h, w, nin, cfg in (("gru16", 92, 156, 3, 4), ("gru32", 46, 78, 2, 4)):
        a = acts(nin, h, w)
        hs = torch.tanh(torch.randn(1, 128, h, w, device=DEV))
        g0, g1, g2 = (torch.randn(1, 128, h, w, device=DEV) for _ in range(3))
        r2, h2 = c8.ActC8(1, 128, h, w, DEV), c8.ActC8(1, 128, h, w, DEV)
        lz, lq = mk(128 * nin, 256), mk(128 * nin, 128)
        zz = c8.gate_zr(a, lz, g0, g1, hs, rh_c8=r2, cfg=cfg)
        report("conv_c8 %s z|r %d->256 (cfg%d)" % (name, 128 * nin, cfg), timeit(lambda: c8.gate_zr(a, lz, g0, g1, hs, rh_c8=r2, cfg=cfg), n=100), flops=fl(128 * nin, 256, h, w))
        report("conv_c8 %s q %d->128 (cfg%d)" % (name, 128 * nin, cfg), timeit(lambda: c8.gate_out(a, lq, g2, zz, hs, hs, out_c8=h2, cfg=cfg), n=100), flops=fl(128 * nin, 128, h, w))
    # motion encoder: convc2 | convf2 in one launch, enc.conv with the flow tail; flow head conv1 with the fused projection
    cor, flo, cf, mf = c8.ActC8(1, 64, H, W, DEV), c8.ActC8(1, 64, H, W, DEV), c8.ActC8(1, 128, H, W, DEV), c8.ActC8(1, 128, H, W, DEV)
    c8.pack(torch.randn(1, 64, H, W, device=DEV), cor)
    c8.pack(torch.randn(1, 64, H, W, device=DEV), flo)
    c2, f2, enc = mk(64, 64), mk(64, 64), mk(128, 126)
    flow = torch.randn(1, 2, H, W, device=DEV)

    def pair():
        d0 = c8.desc([cor], c2, relu=True, out_c8=cf, out_c8_ch0=0)
        d1 = c8.desc([flo], f2, relu=True, out_c8=cf, out_c8_ch0=64)
        c8.launch_pair(d0, d1, flow, 3)
    report("conv_c8 convc2 | convf2 64->64 x2, one launch (cfg3)", timeit(pair, n=100), flops=2 * fl(64, 64, H, W))
    report("conv_c8 encoder.conv 128->126 + flow tail (cfg3)", timeit(lambda: c8.conv2d_c8([cf], enc, relu=True, out_c8=mf, tail=flow, cfg=3), n=100), flops=fl(128, 126, H, W))
    h1, h2l = mk(128, 256), mk(256, 2)
    from dkt_stereo_amd.update import _leading_outputs
    tgt = torch.zeros(1, 1, H, W, device=DEV)
    report("conv_c8 flow_head.conv1 128->256 + fused conv2 (cfg2) + head_finish", timeit(lambda: c8.head([hc], h1, _leading_outputs(h2l, 1), tgt, cfg=2), n=100), flops=fl(128, 256, H, W))
    # producers
    n0, n1 = torch.randn(1, 128, H, W, device=DEV), torch.randn(1, 128, 92, 156, device=DEV)
    p0, u1 = c8.ActC8(1, 128, 92, 156, DEV), c8.ActC8(1, 128, H, W, DEV)
    report("pool2x -> C8S 128ch 184x312", timeit(lambda: c8.pool2x_c8(n0, p0), n=200), bytes_=n0.numel() * 4 + 92 * 156 * 128 * 4)
    report("interp -> C8S 128ch 92x156 -> 184x312", timeit(lambda: c8.interp_c8(n1, u1), n=200), bytes_=n1.numel() * 4 + n0.numel() * 4)
    st7 = torch.nn.Conv2d(2, 64, 7, padding=3).to(DEV)
    report("stem7 convf1 2->64 7x7 -> C8S", timeit(lambda: c8.stem7_c8(flow, st7, flo), n=200), bytes_=flow.numel() * 4 + 64 * H * W * 4)
    # full-resolution encoder layer (opt-in path)
    Hf, Wf = 736, 1248
    xf = c8.pack(torch.randn(1, 64, Hf, Wf, device=DEV))
    yf = c8.ActC8(1, 64, Hf, Wf, DEV)
    e64 = mk(64, 64)
    report("conv_c8 encoder 64->64 @736x1248 -> C8S (cfg3)", timeit(lambda: c8.conv2d_c8([xf], e64, relu=True, out_c8=yf, cfg=3), n=30), flops=fl(64, 64, Hf, Wf))
    del xf, yf
    # lookup fused with convc1 writing C8S (block form)
    c1 = torch.nn.Conv2d(36, 64, 1).to(DEV)
    for B in (1, 8):
        f1, f2_ = (torch.randn(B, 256, H, W, device=DEV) for _ in range(2))
        blk = CorrBlock1D(f1, f2_, num_levels=4, radius=4)
        coords = torch.zeros(B, 2, H, W, device=DEV)
        xs = torch.arange(W, device=DEV).float().view(1, 1, W)
        coords[:, 0] = xs - 20.3 - 3.0 * torch.sin(torch.arange(H, device=DEV).float() / 9.0).view(1, H, 1)
        dst = c8.ActC8(B, 64, H, W, DEV)
        report("lookup + convc1 -> C8S B=%d smooth" % B, timeit(lambda: blk.lookup_conv1x1(coords, c1, out_c8=dst), n=200), bytes_=B * H * W * 420)
        # round 4: the motion encoder's front as one launch (coordinate update + that lookup / 1x1 layer + the 7x7 stem)
        hcb = c8.pack(torch.tanh(torch.randn(B, 128, H, W, device=DEV)))
        planes, n_co = c8.head_planes([hcb], h1, _leading_outputs(h2l, 1), cfg=2)
        x_old, x_new, x0 = coords[:, :1].clone(), torch.empty(B, 1, H, W, device=DEV), coords[:, :1].clone()
        fl2, flo2 = torch.zeros(B, 2, H, W, device=DEV), c8.ActC8(B, 64, H, W, DEV)
        report("motion front B=%d: finish + lookup + convc1 + stem7 -> C8S, one launch" % B,
               timeit(lambda: c8.motion_front(blk, planes, n_co, _leading_outputs(h2l, 1).bias, x_old, x_new, x0, fl2, c1, dst, st7, flo2), n=200),
               bytes_=B * H * W * (420 + 18 * 4 + 12 + 256))
        del blk, f1, f2_, dst
    u2, p1 = c8.ActC8(1, 128, 92, 156, DEV), c8.ActC8(1, 128, 46, 78, DEV)
    n2 = torch.randn(1, 128, 46, 78, device=DEV)
    report("pool2x(1/4) | interp(1/16 -> 1/8) -> C8S, one launch", timeit(lambda: c8.resample_pair_c8(("pool", n0, p0), ("interp", n2, u2)), n=200),
           bytes_=(n0.numel() + n2.numel()) * 4 + 2 * 92 * 156 * 128 * 4)
    report("interp(1/8 -> 1/4) | pool2x(1/8) -> C8S, one launch", timeit(lambda: c8.resample_pair_c8(("interp", n1, u1), ("pool", n1, p1)), n=200),
           bytes_=2 * n1.numel() * 4 + (n0.numel() + 46 * 78 * 128) * 4)
    # group-wise correlation: exact VALU kernel vs the banded MFMA product
    for name, C, G_, h, w in (("IGEV 96ch G=8", 96, 8, 184, 312), ("GwcNet 320ch G=40", 320, 40, 136, 240)):
        a, b = (torch.randn(1, C, h, w, device=DEV) for _ in range(2))
        by = 2 * a.numel() * 4 + G_ * 48 * h * w * 4
        for mode in ("exact", "mfma"):
            with sm.gwc_mode(mode):
                report("gwc_volume %s D=48 [%s]" % (name, mode), timeit(lambda: sm.build_gwc_volume(a, b, 48, G_), n=50), bytes_=by,
                       flops=2.0 * C * 48 * h * w)


def main():
    which = sys.argv[1:] or ["lookup", "build", "gates", "conv", "e2e", "autocast", "volumes"]
    fns = dict(lookup=bench_lookup, build=bench_build, gates=bench_gates, conv=bench_conv, e2e=bench_e2e,
               autocast=bench_autocast, volumes=bench_volumes, ablate=bench_ablate, next=bench_next, c8=bench_c8)
    for w in which:
        print("== %s ==" % w, flush=True)
        try:
            fns[w]()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print("!! %s failed: %s" % (w, e))
    out = os.path.join(ROOT, "gpurun_out", "bench_kernels.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(RESULTS, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
