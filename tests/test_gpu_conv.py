"""-m gpu: the split-fp16 implicit-GEMM convolution (dkt_conv2d_f16s) against an fp64
reference of nn.Conv2d, alone and inside the update operator / the full
RAFT-Stereo loop.  Tolerances, relative to max|y|:
   passes=3 (w_hi*x_hi + w_lo*x_hi + w_hi*x_lo):  2e-6   (fp32-class)
   passes=2 (activations rounded to fp16)       :  1e-3
   passes=1 (plain fp16 operands)               :  3e-3
End to end the passes=3 backend must meet north_star's 1e-3 max-abs on the final
disparity against the reference's outputs (tests/golden)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases
import _synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL = {"f16x3": 2e-6, "f16x2": 1e-3, "f16": 3e-3}


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(autouse=True)
def _restore_backend():
    from dkt_stereo_amd import conv
    prev = conv.get_backend()
    yield
    conv.set_backend(prev)


SHAPES = [
    # name, B, src channels, Cout, H, W, k, relu
    ("k3_small", 2, [16], 40, 9, 37, 3, False),
    ("k3_cat3", 1, [128, 128, 128], 256, 10, 45, 3, False),      # gru08 z|r shape
    ("k3_cat_odd", 1, [7, 33, 64], 126, 6, 33, 3, True),         # ragged sources, Cout not /64
    ("k3_narrow", 2, [64], 64, 17, 70, 3, True),                 # WM=1 layout (encoder 64->64)
    ("k3_cout2", 1, [256], 2, 8, 40, 3, False),                  # flow head conv2
    ("k1_corr", 2, [36], 64, 7, 50, 1, True),                    # encoder convc1
    ("k1_mask", 1, [256], 144, 5, 33, 1, False),                 # mask head 1x1
    ("k3_tiny", 1, [3], 5, 2, 3, 3, False),                      # smaller than one tile
]


@pytest.mark.parametrize("backend", ["f16x3", "f16x2", "f16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
@torch.no_grad()
def test_conv_vs_fp64(shape, backend):
    from dkt_stereo_amd import conv
    name, B, chans, cout, H, W, k, relu = shape
    cin = sum(chans)
    layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
    bnd = 1.0 / np.sqrt(cin * k * k)
    layer.weight.data = G(_synth.uniform((cout, cin, k, k), -bnd, bnd, 90, name, "w"))
    layer.bias.data = G(_synth.uniform((cout,), -bnd, bnd, 90, name, "b"))
    layer.to(DEV)
    xs = [G(_synth.normal((B, c, H, W), 90, name, "x%d" % i, scale=1.5)) for i, c in enumerate(chans)]
    ref = F.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), padding=k // 2)
    if relu:
        ref = ref.clamp_min(0)
    conv.set_backend(backend)
    got = conv.conv2d(xs if len(xs) > 1 else xs[0], layer, relu=relu)
    assert got.shape == ref.shape and got.dtype == torch.float32
    err = float((got.double() - ref).abs().max())
    scale = float(ref.abs().max())
    print("%s %s: max err %.3e (rel %.2e)" % (name, backend, err, err / scale))
    assert err <= REL[backend] * scale
    # strided batch operand (a channel slice of a wider tensor) is read in place
    wide = torch.zeros(B, chans[0] + 3, H, W, device=DEV)
    wide[:, 1:1 + chans[0]] = xs[0]
    got2 = conv.conv2d([wide[:, 1:1 + chans[0]]] + xs[1:], layer, relu=relu)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("shape", [("head", 1, 256, 2, 23, 70, 3, False), ("disp", 2, 256, 1, 9, 33, 3, False),
                                   ("stem", 1, 2, 64, 20, 45, 7, True), ("dstem", 2, 1, 64, 11, 34, 7, True),
                                   ("c4", 1, 40, 4, 8, 8, 3, True)], ids=lambda s: s[0])
@torch.no_grad()
def test_direct_conv_vs_fp64(shape):
    """dkt_conv2d_direct (exact fp32 FMAs): flow/disp head (Cout <= 4) and 7x7 stems (Cin <= 4)."""
    from dkt_stereo_amd import conv
    name, B, cin, cout, H, W, k, relu = shape
    conv.set_backend("f16x3")
    torch.manual_seed(1234)
    layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(DEV)
    assert conv.direct_eligible(layer) == (k == 7)
    x = G(_synth.normal((B, cin, H, W), 95, name, scale=2.0))
    ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=k // 2)
    ref = ref.clamp_min(0) if relu else ref
    got = conv._conv2d_direct(x, layer, relu, None)
    # sequential fp32 FMAs over K = cin*k*k terms: round-off grows like sqrt(K) * 2^-24
    tol = max(2e-6, 1.5e-7 * (cin * k * k) ** 0.5)
    assert float((got.double() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [("flow", 1, 2, 64, 20, 45), ("disp", 2, 1, 64, 11, 34), ("rgb", 1, 3, 64, 37, 70),
                                   ("c4_co100", 1, 4, 100, 5, 33), ("tiny", 1, 2, 64, 1, 1)], ids=lambda s: s[0])
@torch.no_grad()
def test_stem7_vs_fp64(shape):
    """dkt_conv2d_stem7 (7x7, Cin <= 4, split-fp16 MFMA with K over the taps) vs an fp64 convolution and
    the exact-fp32 direct kernel."""
    from dkt_stereo_amd import conv
    name, B, cin, cout, H, W = shape
    conv.set_backend("f16x3")
    torch.manual_seed(4321)
    layer = torch.nn.Conv2d(cin, cout, 7, padding=3).to(DEV)
    assert conv.direct_eligible(layer)
    x = G(_synth.normal((B, cin, H, W), 97, name, scale=3.0))
    for relu in (False, True):
        ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=3)
        ref = ref.clamp_min(0) if relu else ref
        got = conv.conv2d(x, layer, relu=relu)
        tol = 3e-6 * max(1.0, float(ref.abs().max()))
        assert got.shape == ref.shape and float((got.double() - ref).abs().max()) <= tol
        assert float((got - conv._conv2d_direct(x, layer, relu, None)).abs().max()) <= 2 * tol
    layer.weight.mul_(2.0)                                   # cache invalidation through the version counter
    ref2 = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=3)
    assert float((conv.conv2d(x, layer).double() - ref2).abs().max()) <= 6e-6 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("shape", [("s2_k3", 2, 64, 96, 23, 70, 3), ("s2_k3_odd", 1, 96, 128, 9, 33, 3),
                                   ("s2_k1", 2, 64, 96, 12, 41, 1), ("s2_k3_wide", 1, 32, 160, 8, 64, 3),
                                   ("s2_k3_tiny", 1, 8, 24, 1, 1, 3), ("s2_k1_big", 1, 128, 128, 46, 78, 1)],
                         ids=lambda s: s[0])
@torch.no_grad()
def test_strided_conv_vs_fp64(shape):
    """dkt_conv2d_f16s_strided (stride 2, padding K/2): the encoders' down-sampling convolutions
    (core/extractor.py:16,34), incl. odd sizes and ReLU, against an fp64 convolution."""
    from dkt_stereo_amd import conv
    name, B, cin, cout, H, W, k = shape
    conv.set_backend("f16x3")
    torch.manual_seed(77)
    layer = torch.nn.Conv2d(cin, cout, k, stride=2, padding=k // 2).to(DEV)
    assert conv.hip_eligible(layer)
    x = G(_synth.normal((B, cin, H, W), 96, name, scale=1.5))
    for relu in (False, True):
        ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), stride=2, padding=k // 2)
        ref = ref.clamp_min(0) if relu else ref
        got = conv.conv2d(x, layer, relu=relu)
        assert got.shape == ref.shape
        assert float((got.double() - ref).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max()))


@torch.no_grad()
def test_conv_weight_cache_invalidation():
    from dkt_stereo_amd import conv
    conv.set_backend("f16x3")
    layer = torch.nn.Conv2d(32, 64, 3, padding=1).to(DEV)
    x = G(_synth.normal((1, 32, 8, 40), 91, "x"))
    a = conv.conv2d(x, layer)
    # in-place writes to the Parameter (what load_state_dict / optimizers do) bump its version
    # counter; writes through `.data` do not -- conv.clear_weight_cache() covers those
    layer.weight.mul_(2.0)
    layer.bias.zero_()
    b = conv.conv2d(x, layer)
    ref = F.conv2d(x.double(), layer.weight.double(), None, padding=1)
    assert float((b.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert not torch.equal(a, b)


@pytest.mark.parametrize("name", list(_cases.UPDATE_CASES))
@torch.no_grad()
def test_update_block_f16x3(name, golden):
    from dkt_stereo_amd import conv
    from test_gpu_parity import _make_block, maxabs
    conv.set_backend("f16x3")
    c = _cases.UPDATE_CASES[name]
    blk, _ = _make_block(c)
    net, inp, corr, flow = _cases.update_inputs(c)
    n = c["n"]
    gnet = [G(x) for x in net]
    ginp = [list(G(x).split(128, dim=1)) for x in inp]
    kw = dict(iter16=(n == 3), iter08=(n >= 2)) if c["igev"] else dict(iter32=(n == 3), iter16=(n >= 2))
    rnet, rmask, rdelta = blk(gnet, ginp, G(corr), G(flow), **kw)
    g = golden("update")
    for i in range(3):
        assert maxabs(rnet[i], g["%s/net%d" % (name, i)]) <= 1e-5
    assert maxabs(rmask[:, :, ::2, ::2], g[name + "/mask"]) <= 5e-5
    assert maxabs(rdelta, g[name + "/delta"]) <= 5e-5


@pytest.mark.parametrize("name", list(_cases.E2E_CASES))
@torch.no_grad()
def test_raft_stereo_end_to_end_f16x3(name, golden):
    from dkt_stereo_amd import conv
    from test_gpu_parity import _raft, maxabs
    conv.set_backend("f16x3")
    c = _cases.E2E_CASES[name]
    model, _ = _raft()
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    lo, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
    g = golden("raft_e2e")
    s = int(g[name + "/stride"])
    d_up = maxabs(up[:, :, ::s, ::s], g[name + "/flow_up"])
    epe = float(np.mean(np.abs(up[:, :, ::s, ::s].cpu().numpy() - g[name + "/flow_up"])))
    print("%s f16x3: max|d_up| %.3e EPE %.3e" % (name, d_up, epe))
    assert d_up <= 1e-3 and epe <= 1e-3


@pytest.mark.parametrize("backend", ["f16x2", "f16"])
@torch.no_grad()
def test_reduced_pass_backends_report_their_deviation(backend, golden):
    """The cheaper modes are not parity paths; this records how far they drift on
    the 32-iteration fixture and only guards against gross breakage."""
    from dkt_stereo_amd import conv
    from test_gpu_parity import _raft, maxabs
    conv.set_backend(backend)
    c = _cases.E2E_CASES["256x512_it32"]
    model, _ = _raft()
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    _, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
    g = golden("raft_e2e")
    s = int(g["256x512_it32/stride"])
    d = maxabs(up[:, :, ::s, ::s], g["256x512_it32/flow_up"])
    epe = float(np.mean(np.abs(up[:, :, ::s, ::s].cpu().numpy() - g["256x512_it32/flow_up"])))
    print("256x512_it32 %s: max|d_up| %.3e EPE %.3e" % (backend, d, epe))
    assert d <= 0.5


@torch.no_grad()
def test_hip_graph_replay_matches_eager():
    """The captured-iteration path (RAFTStereo.use_hip_graph) must give exactly the eager
    result on identical encoder outputs: on the first call (capture), on a later call with
    another pair (replay only) and after a shape change (fresh capture).  (The encoders are
    run once per pair and shared: vendor convolutions may pick another algorithm per call.)"""
    from test_gpu_parity import _raft, maxabs
    model, _ = _raft()
    model.use_c8 = False                  # the round-2 loop: its captured iteration IS the plain loop's arithmetic
    model._graph_state = None
    for (h, w, iters, seed, shift) in ((128, 256, 7, 0, 12), (128, 256, 7, 5, 30), (64, 128, 5, 1, 12)):
        i1, i2 = _synth.image_pair(seed, 1, h, w, shift)
        fmap1, fmap2, net, inp = model.encode(G(i1), G(i2))
        model.use_hip_graph = False
        lo_e, up_e = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)
        model.use_hip_graph = True
        lo_g, up_g = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)
        assert torch.equal(lo_g, lo_e) and torch.equal(up_g, up_e), (h, w, seed)
    assert model._graph_state is not None and model._graph_state["graph"] is not None
    # the default loop (loop_c8, every size since round 4): replay == the same units launched eagerly, bit for bit; the plain
    # loop on the round-2 kernels is another summation order of the same split-fp16 products
    model, _ = _raft()
    for (h, w, iters, seed, shift) in ((128, 256, 7, 0, 12), (128, 256, 7, 5, 30), (64, 128, 5, 1, 12)):
        i1, i2 = _synth.image_pair(seed, 1, h, w, shift)
        fmap1, fmap2, net, inp = model.encode(G(i1), G(i2))
        net = [t.clone() for t in net]
        inp = [[t.clone() for t in sc] for sc in inp]
        model.use_hip_graph = False
        lo_p, up_p = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)
        model.use_hip_graph = True
        lo_g, up_g = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)      # first call of a shape: eager unit + capture + replay
        lo_r, up_r = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)      # replay only
        assert model._graph_state.get("c8") is not None and model._graph_state["c8"].graph
        model.c8_eager = True
        try:
            lo_e, up_e = model.iterate(fmap1, fmap2, [t.clone() for t in net], inp, iters)
        finally:
            model.c8_eager = False
        assert torch.equal(lo_g, lo_r) and torch.equal(up_g, up_r) and torch.equal(lo_g, lo_e) and torch.equal(up_g, up_e), (h, w, seed)
        assert maxabs(up_g, up_p) <= 2e-4 * max(1.0, float(up_p.abs().max())), (h, w, seed)


@torch.no_grad()
def test_streaming_helpers_vs_torch():
    """dkt_pool2x / dkt_interp_bilinear (core/update.py:87-95) and the encoder glue
    (instance norm [+ReLU], add+ReLU) against the torch ops they replace."""
    from dkt_stereo_amd.extractor import add_relu, norm_act
    from dkt_stereo_amd.update import interp, pool2x
    for (B, C, H, W) in ((1, 128, 184, 312), (2, 5, 7, 9), (1, 3, 46, 78), (1, 2, 5, 6)):
        x = G(_synth.normal((B, C, H, W), 92, "x%d" % H))
        want = F.avg_pool2d(x, 3, stride=2, padding=1)
        got = pool2x(x)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-6
        dest = torch.empty(B, C, 2 * H, 2 * W, device=DEV)
        want = F.interpolate(x, dest.shape[2:], mode="bilinear", align_corners=True)
        got = interp(x, dest)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6
        odd = torch.empty(B, C, 2 * H + 1, 2 * W - 1, device=DEV)
        assert float((interp(x, odd) - F.interpolate(x, odd.shape[2:], mode="bilinear", align_corners=True)).abs().max()) <= 2e-6
    for (B, C, H, W) in ((2, 3, 10, 16), (1, 2, 9, 640), (1, 4, 1, 8), (3, 1, 6, 24)):      # wide / tiny / batched planes
        x = G(_synth.normal((B, C, H, W), 96, "p%d" % H))
        want = F.avg_pool2d(x, 3, stride=2, padding=1)
        got = pool2x(x)
        assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-6
    # the LDS-staged up-sampling form (16-byte aligned rows): the loop's 1/8 -> 1/4 resize, ragged row blocks, non-2x scales
    for (B, C, H, W, Ho, Wo) in ((1, 16, 92, 156, 184, 312), (2, 3, 20, 24, 37, 44), (1, 2, 8, 8, 8, 8), (1, 5, 33, 640, 70, 1000)):
        x = G(_synth.normal((B, C, H, W), 95, "u%d" % H))
        want = F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=True)
        got = interp(x, torch.empty(B, C, Ho, Wo, device=DEV))
        assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6
    from dkt_stereo_amd.extractor import norm_add_relu
    inorm = torch.nn.InstanceNorm2d(8)
    for (B, C, H, W) in ((2, 8, 33, 47), (1, 8, 64, 128), (1, 8, 3, 5)):
        xa = G(_synth.normal((B, C, H, W), 94, "ja%d" % H))
        xc = G(_synth.normal((B, C, H, W), 94, "jc%d" % H, scale=3.0)) + 0.7
        want = torch.relu(xa.double() + torch.relu(inorm(xc.double()))).float()
        assert float((norm_add_relu(inorm, xa, xc) - want).abs().max()) <= 3e-6
    for (B, C, H, W) in ((2, 8, 33, 47), (1, 8, 64, 128), (1, 8, 3, 5)):
        x = G(_synth.normal((B, C, H, W), 93, "in%d" % H, scale=3.0)) + 1.5
        for relu in (False, True):
            want = inorm(x.double()).float()
            want = want.clamp_min(0) if relu else want
            assert float((norm_act(inorm, x, relu) - want).abs().max()) <= 5e-6
    a, b = (G(_synth.normal((2, 7, 9, 11), 94, n)) for n in ("a", "b"))
    assert torch.equal(add_relu(a, b), F.relu(a + b))
    bn = torch.nn.BatchNorm2d(7).to(DEV).eval()       # every other norm stays on torch
    assert torch.equal(norm_act(bn, a, True), F.relu(bn(a)))
