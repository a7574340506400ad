"""-m gpu: round 4 -- one ConvGRU step per launch (csrc/gru_c8.hip, core/update.py:23-32).

Oracles: torch's fp64 ConvGRU arithmetic on the same operands (bound 4e-6 relative: the split-fp16 class, with the context
terms as the accumulators' start value), the two-launch form of round 3 (conv_c8 epilogues 1 + 2: the same arithmetic in
another summation order) and, through the full model, the reference fixtures (tests/test_gpu_round3.py runs the loop that
now takes this launch).  The inter-tile hand-off (write-through r*h, per-tile flags, one agent acquire) is exercised with
batches that give every block several tiles, launch after launch on the same flag words, and must be bit-reproducible.
"""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max())


def _c8():
    from dkt_stereo_amd import conv_c8
    return conv_c8


def _make(B, H, W, xch, seed, ctx_scale=1.0):
    from dkt_stereo_amd.update import ConvGRU
    torch.manual_seed(seed)
    gru = ConvGRU(128, sum(xch)).to(DEV)
    h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    xs = [torch.randn(B, c, H, W, device=DEV) for c in xch]
    cz, cr, cq = (torch.randn(B, 128, H, W, device=DEV) * ctx_scale for _ in range(3))
    return gru, h, xs, cz, cr, cq


def _ref64(gru, h, xs, cz, cr, cq):
    """core/update.py:23-32 in fp64."""
    p = {k: v.double() for k, v in gru.state_dict().items()}
    h, cz, cr, cq = h.double(), cz.double(), cr.double(), cq.double()
    x = torch.cat([t.double() for t in xs], 1)
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(F.conv2d(hx, p["convz.weight"], p["convz.bias"], padding=1) + cz)
    r = torch.sigmoid(F.conv2d(hx, p["convr.weight"], p["convr.bias"], padding=1) + cr)
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), p["convq.weight"], p["convq.bias"], padding=1) + cq)
    return (1 - z) * h + z * q


class _State:
    def __init__(self, gru, h, xs, cz, cr, cq):
        c8 = _c8()
        B, _, H, W = h.shape
        self.gru, self.h, self.cz, self.cr, self.cq = gru, h.clone(), cz, cr, cq
        self.hc8 = c8.pack(self.h)
        self.xs = [c8.pack(x) for x in xs]
        self.rh = c8.ActC8(B, 128, H, W, DEV)
        self.flags = c8.gru_flags(B, H, W, DEV)

    def desc(self):
        return _c8().gru_desc(self.gru, self.hc8, self.xs, self.rh, self.cz, self.cr, self.cq, self.h, self.flags)

    def two_launch(self):
        c8 = _c8()
        z = c8.gate_zr([self.hc8, *self.xs], self.gru._merged_zr(), self.cz, self.cr, self.h, rh_c8=self.rh, cfg=1)
        c8.gate_out([self.rh, *self.xs], self.gru.convq, self.cq, z, self.h, self.h, out_c8=self.hc8, cfg=2)


@torch.no_grad()
@pytest.mark.parametrize("B,H,W,xch", [(1, 96, 160, [128, 128]),      # 60 tiles, one per block
                                       (1, 23, 39, [128]),            # the coarsest GRU of cfg2: 6 tiles, ragged on both axes
                                       (1, 50, 70, [64, 40, 24]),     # three operands, channel counts that pad to 16
                                       (2, 184, 312, [128, 128]),     # 460 tiles on 256 CUs: two tiles per block, neighbours across rounds
                                       (5, 96, 160, [128])])          # 300 tiles
def test_fused_gru_step_matches_fp64_and_the_two_launch_form(B, H, W, xch):
    c8 = _c8()
    args = _make(B, H, W, xch, seed=B * 1000 + H)
    a, b = _State(*args), _State(*args)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    h_ref = args[1]
    for step in range(3):                        # the same flag words launch after launch; the state evolves
        want = _ref64(args[0], h_ref, *args[2:])
        assert c8.gru_launch(a.desc(), err=err), "the device declined a launch it must be able to hold"
        b.two_launch()
        assert _rel(a.h, want) <= 4e-6 and _rel(b.h, want) <= 4e-6, (step, _rel(a.h, want), _rel(b.h, want))
        # the C8S twin is the split of the fp32 state; border and padding stay zero
        assert torch.equal(a.hc8.t, c8.pack(a.h).t)
        assert float(a.hc8.t[:, :, :, 0].abs().max()) == 0 and float(a.hc8.t[:, :, :, :, 0].abs().max()) == 0
        assert float(a.rh.t[:, :, :, H + 1:].abs().max()) == 0 and float(a.rh.t[:, :, :, :, W + 1:].abs().max()) == 0
        h_ref = want.float()
        a.h.copy_(h_ref); b.h.copy_(h_ref)       # both forms continue from the oracle's state
        c8.pack(a.h, a.hc8); c8.pack(b.h, b.hc8)
    assert int(err.item()) == 0
    assert int(a.flags.min()) == 3 and int(a.flags.max()) == 3


@torch.no_grad()
def test_fused_gru_pair_equals_single_launches_and_is_reproducible():
    """Two ConvGRU steps in one launch (the finest level with the coarsest riding along, as the loop issues them) are
    bit-identical to the two single launches; twelve launches from the same state give one result."""
    c8 = _c8()
    big, small = _make(1, 184, 312, [128, 128], 11), _make(1, 23, 39, [128], 12)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    sa, sb = _State(*big), _State(*small)
    assert c8.gru_launch(sa.desc(), sb.desc(), err=err)
    ra, rb = _State(*big), _State(*small)
    assert c8.gru_launch(ra.desc(), err=err) and c8.gru_launch(rb.desc(), err=err)
    assert torch.equal(sa.h, ra.h) and torch.equal(sb.h, rb.h) and torch.equal(sa.hc8.t, ra.hc8.t) and torch.equal(sb.hc8.t, rb.hc8.t)
    for _ in range(12):
        ta, tb = _State(*big), _State(*small)
        c8.gru_launch(ta.desc(), tb.desc(), err=err)
        assert torch.equal(ta.h, sa.h) and torch.equal(tb.h, sb.h) and torch.equal(ta.rh.t, sa.rh.t)
    assert int(err.item()) == 0


@torch.no_grad()
def test_fused_gru_hand_off_under_uneven_load():
    """The neighbour hand-off with the device busy elsewhere: a streaming kernel on a second stream keeps some CUs' memory
    queues loaded while 40 dependent steps run; every step must equal the undisturbed run bit for bit (a stale r*h halo or
    an early state update would change the result)."""
    c8 = _c8()
    args = _make(2, 184, 312, [128, 128], 21)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    quiet = _State(*args)
    want = []
    for _ in range(40):
        c8.gru_launch(quiet.desc(), err=err)
        want.append(quiet.h.clone())
    torch.cuda.synchronize()
    noisy = _State(*args)
    side = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=DEV)
    stop = torch.cuda.Event()
    got = []
    with torch.cuda.stream(side):
        for _ in range(60):
            junk.mul_(1.0001)
        stop.record()
    for _ in range(40):
        c8.gru_launch(noisy.desc(), err=err)
        got.append(noisy.h.clone())
    stop.synchronize()
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(got, want))
    assert int(err.item()) == 0


@torch.no_grad()
def test_fused_gru_context_terms_of_any_size():
    """The accumulators start at (bias + context) / scale: large and tiny context terms keep the fp64 bound."""
    c8 = _c8()
    for scale in (1e-3, 30.0):
        args = _make(1, 48, 96, [128], 31, ctx_scale=scale)
        s = _State(*args)
        assert c8.gru_launch(s.desc())
        assert _rel(s.h, _ref64(*args)) <= 4e-6


@torch.no_grad()
def test_fused_gru_argument_errors():
    import ctypes
    c8 = _c8()
    from dkt_stereo_amd import _ffi
    args = _make(1, 32, 64, [128], 41)
    s = _State(*args)
    d = s.desc()
    d.hidden = 96
    assert _ffi.lib().dkt_gru_c8(ctypes.byref(d), None, 0, None) == _ffi.E_UNSUPPORTED
    d = s.desc()
    d.flags = None
    assert _ffi.lib().dkt_gru_c8(ctypes.byref(d), None, 0, None) < 0
    with pytest.raises(ValueError):
        c8.gru_desc(s.gru, s.hc8, s.xs, c8.ActC8(1, 128, 16, 64, DEV), s.cz, s.cr, s.cq, s.h, s.flags)


def _rescaled_state_dict(sd, k):
    """The same function with the update block's unbounded activations (correlation / flow features, their 64-channel
    successors, the motion features, the flow head's hidden tensor) multiplied by 2^k: every layer between two of them is
    positively homogeneous (conv + ReLU), so scaling a layer's weights and bias by 2^k and dividing its consumers' weights
    for those channels by 2^k changes nothing -- exactly nothing in fp32, powers of two commute with rounding."""
    s = 2.0 ** k
    sd = {n: v.clone() for n, v in sd.items()}
    enc = "update_block.encoder."
    for name in ("convc1", "convf1"):
        sd[enc + name + ".weight"] *= s
        sd[enc + name + ".bias"] *= s
    for name in ("convc2", "convf2", "conv"):          # input and output both scaled: weights unchanged, bias scaled
        sd[enc + name + ".bias"] *= s
    for g in ("convz", "convr", "convq"):              # gru08 reads [h (128) | motion features (126) + flow (2) | interp (128)]
        sd["update_block.gru08.%s.weight" % g][:, 128:254] /= s
    sd["update_block.flow_head.conv1.weight"] *= s
    sd["update_block.flow_head.conv1.bias"] *= s
    sd["update_block.flow_head.conv2.weight"] /= s
    return sd


@torch.no_grad()
@pytest.mark.parametrize("k", [-10, 10])
def test_default_path_holds_the_bound_when_activations_sit_at_1e_3_or_1e3(k, golden):
    """the C8S format has 22 significant bits only inside fp16's normal range.  With the update block's
    activations at 2^-10 (~1e-3) or 2^10 (~1e3) of their usual size the DEFAULT path (no conv.calibrate()) must still be within
    1e-3 of the reference's output on the benchmark workload -- the reference's own fixture applies unchanged, the rescaling is
    exact in its fp32 arithmetic."""
    import math
    import _cases
    import _synth
    from test_gpu_parity import G, _raft, maxabs
    c = _cases.E2E_CASES["736x1248_it32"]
    g = golden("raft_e2e")
    st = int(g["736x1248_it32/stride"])
    model, sd = _raft()
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"]))
    _, up = model(i1, i2, iters=32, test_mode=True)
    base = maxabs(up[:, :, ::st, ::st], g["736x1248_it32/flow_up"])
    lp = model._graph_state["c8"]
    assert lp.calibrated
    e0 = {n: math.log2(getattr(lp, n).scale) for n in ("cor", "flo", "cf", "mf")}
    t0 = lp.mf.tail_scale
    model.load_state_dict(_rescaled_state_dict(sd, k), strict=True)
    _, up = model(i1, i2, iters=32, test_mode=True)
    d = maxabs(up[:, :, ::st, ::st], g["736x1248_it32/flow_up"])
    print("activations x 2^%d: %.3e from the reference fixture (unscaled weights: %.3e)" % (k, d, base))
    assert d <= 1e-3
    lp = model._graph_state["c8"]
    e1 = {n: math.log2(getattr(lp, n).scale) for n in e0}
    assert all(e1[n] == e0[n] - k for n in e0), (e0, e1)          # the scales followed the activations
    assert lp.mf.tail_scale == t0 and lp.mf.tail_scale != lp.mf.scale          # the flow channels kept theirs


@torch.no_grad()
@pytest.mark.parametrize("shape", [(2, 64, 64, 96, 160, 1, 3), (2, 64, 96, 96, 160, 2, 3), (1, 96, 128, 47, 81, 2, 3),
                                   (2, 64, 96, 90, 130, 2, 1), (1, 128, 128, 40, 72, 1, 3), (3, 32, 40, 33, 70, 1, 3),
                                   (2, 96, 96, 93, 157, 1, 3), (1, 96, 80, 120, 200, 1, 3), (2, 64, 96, 264, 544, 2, 3),
                                   (2, 96, 128, 264, 544, 2, 3)])      # the 96-channel wave tile (three blocks per wave)
def test_conv_epilogue_statistics_match_the_statistics_pass(shape):
    """dkt_conv_desc.stats_ws: the (mean, 1/std) an InstanceNorm2d needs of a convolution's output, accumulated in the
    convolution's epilogue, against dkt_instance_norm_stats on the written output and against torch in fp64 -- every tile
    shape the encoders use (stride 1 / 2, 1x1 / 3x3, ragged sizes, channel counts off the 32-channel grid)."""
    from dkt_stereo_amd import conv, extractor
    B, cin, cout, H, W, stride, k = shape
    torch.manual_seed(cout + H)
    layer = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).to(DEV)
    x = torch.randn(B, cin, H, W, device=DEV) * 2 + 0.5
    norm = torch.nn.InstanceNorm2d(cout)
    with conv.use_backend("f16x3"):
        want_out = conv.conv2d(x, layer)
        out, st = conv.conv2d_stats(x, layer)
        assert torch.equal(out, want_out)
        got = extractor.instance_norm_params(norm, out, st)
        ref = extractor.instance_norm_params(norm, out)
    o64 = out.double()
    mean = o64.mean((2, 3)).reshape(-1)
    invstd = 1.0 / torch.sqrt(o64.var((2, 3), unbiased=False).reshape(-1) + norm.eps)
    scale = float(o64.abs().max())
    assert float((got[:, 0].double() - mean).abs().max()) <= 2e-6 * scale
    assert float((got[:, 1].double() / invstd - 1).abs().max()) <= 2e-6
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,H,W", [(1, 46, 78), (2, 33, 37), (1, 92, 156)])
def test_resample_pair_equals_single_launches(B, H, W):
    """dkt_resample_pair_c8 (pool2x | interp in one launch, core/update.py:87-95) writes the bytes of the two single launches;
    jobs of different sizes, either order, a scaled destination; shape errors are refused."""
    c8 = _c8()
    torch.manual_seed(H)
    fine = torch.randn(B, 128, 2 * H, 2 * W - 1, device=DEV)
    coarse = torch.randn(B, 128, (H + 1) // 2, (W + 1) // 2, device=DEV)
    Hm, Wm = (fine.shape[2] - 1) // 2 + 1, (fine.shape[3] - 1) // 2 + 1

    def fresh(scale):
        a, b = c8.ActC8(B, 128, Hm, Wm, DEV), c8.ActC8(B, 128, Hm, Wm, DEV)
        a.scale = b.scale = scale
        return a, b

    for scale in (1.0, 64.0):
        p0, u0 = fresh(scale)
        c8.pool2x_c8(fine, p0)
        c8.interp_c8(coarse, u0)
        for order in (0, 1):
            p1, u1 = fresh(scale)
            jobs = [("pool", fine, p1), ("interp", coarse, u1)]
            c8.resample_pair_c8(*(jobs if order == 0 else jobs[::-1]))
            assert torch.equal(p0.t, p1.t) and torch.equal(u0.t, u1.t)
    with pytest.raises(ValueError):
        c8.resample_pair_c8(("pool", fine, c8.ActC8(B, 128, Hm + 1, Wm, DEV)), ("interp", coarse, u0))


@torch.no_grad()
@pytest.mark.parametrize("B,H,W,radius", [(1, 46, 78, 4), (2, 40, 96, 4), (1, 33, 130, 3)])
def test_motion_front_equals_the_three_launches(B, H, W, radius):
    """dkt_motion_front_c8 == dkt_head_finish, then lookup + convc1 -> C8S, then the 7x7 stem -> C8S, bit for bit
    (raft_stereo.py:165-168, core/corr.py:127-146 + core/update.py:76-77): coordinate, flow, both C8S operands; ragged
    widths (the stem's 32-column and the lookup's 64-pixel tiles both end inside the row), batch strides of a 2-channel
    coordinate tensor, scaled destinations; x_new aliasing x_old is refused."""
    from dkt_stereo_amd.corr import CorrBlock1D
    from dkt_stereo_amd.update import FlowHead, _leading_outputs
    c8 = _c8()
    torch.manual_seed(B * 1000 + W)
    K = 2 * radius + 1
    fh = FlowHead(128, 256, 2).to(DEV)
    h = c8.pack(torch.tanh(torch.randn(B, 128, H, W, device=DEV)))
    f1, f2 = torch.randn(B, 64, H, W, device=DEV), torch.randn(B, 64, H, W, device=DEV)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=radius)
    convc1 = torch.nn.Conv2d(4 * K, 64, 1).to(DEV)
    convf1 = torch.nn.Conv2d(2, 64, 7, padding=3).to(DEV)
    xs = torch.arange(W, device=DEV, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H, device=DEV, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
    coords0 = torch.cat([xs, ys], 1).contiguous()
    start = coords0.clone()
    start[:, :1] -= 20 * torch.rand(B, 1, H, W, device=DEV)
    last = _leading_outputs(fh.conv2, 1)
    planes, n_co = c8.head_planes([h], fh.conv1, last, cfg=2)

    for scale in (1.0, 32.0):
        # the three launches
        c_ref, f_ref = start.clone(), torch.zeros(B, 2, H, W, device=DEV)
        f_ref[:, 1] = 0.25                                   # (the y plane is only read)
        c8.head([h], fh.conv1, last, c_ref[:, :1], diff=(coords0[:, :1], f_ref[:, :1]), cfg=2)
        cor_ref, flo_ref = c8.ActC8(B, 64, H, W, DEV, scale=scale), c8.ActC8(B, 64, H, W, DEV, scale=scale)
        assert blk.lookup_conv1x1(c_ref, convc1, out_c8=cor_ref) is not None
        c8.stem7_c8(f_ref, convf1, flo_ref)
        # one launch
        c_old, c_new, flow = start.clone(), torch.full_like(start, -7.0), torch.zeros(B, 2, H, W, device=DEV)
        flow[:, 1] = 0.25
        cor, flo = c8.ActC8(B, 64, H, W, DEV, scale=scale), c8.ActC8(B, 64, H, W, DEV, scale=scale)
        assert c8.motion_front_supported(blk, type("E", (), dict(convc1=convc1, convf1=convf1)))
        c8.motion_front(blk, planes, n_co, last.bias, c_old[:, :1], c_new[:, :1], coords0[:, :1], flow, convc1, cor, convf1, flo)
        assert torch.equal(c_old, start)                                       # the old coordinate is only read
        assert torch.equal(c_new[:, :1], c_ref[:, :1]) and bool((c_new[:, 1] == -7.0).all())
        assert torch.equal(flow, f_ref)
        assert torch.equal(cor.t, cor_ref.t) and torch.equal(flo.t, flo_ref.t)
    with pytest.raises(ValueError):
        c8.motion_front(blk, planes, n_co, last.bias, c_old[:, :1], c_old[:, :1], coords0[:, :1], flow, convc1, cor, convf1, flo)


@torch.no_grad()
def test_encoder_outputs_in_place_and_prebuilt_volume_change_nothing():
    """From the second forward of a shape on, encode() writes the hidden states / context terms into the captured loop's state
    buffers and rebuilds the loop's correlation block on the feature encoder's stream (RAFTStereo.adopt_encoder_outputs,
    prebuild_corr): same disparity, bit for bit, as a model that keeps private tensors and builds the volume in iterate();
    the aliasing is what it claims to be; another pair through the same state stays right."""
    import _synth
    from test_gpu_parity import G, _raft
    H, W = 384, 640                                  # 96 x 160 at 1/4: the C8S loop (>= 24 000 quarter-resolution pixels is not
    a, _ = _raft()                                   # needed for the aliasing itself -- the round-2 loop shares the state logic)
    b, _ = _raft()
    b.adopt_encoder_outputs = b.prebuild_corr = False
    pairs = [tuple(G(t) for t in _synth.image_pair(s, 1, H, W, 12 + 9 * s)) for s in (0, 1, 2)]
    for k, (i1, i2) in enumerate(pairs + pairs[:1]):
        ra, rb = a(i1, i2, iters=6, test_mode=True), b(i1, i2, iters=6, test_mode=True)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]), k
    st = a._graph_state
    f1, f2, net, inp = a.encode(*pairs[1])
    assert [t.data_ptr() for t in net] == [t.data_ptr() for t in st["net"]]
    assert [t.data_ptr() for sc in inp for t in sc] == [t.data_ptr() for sc in st["inp"] for t in sc]
    assert a._prebuilt[0] is st["corr"] and a._prebuilt[1:] == (f1.data_ptr(), f2.data_ptr())
    n2 = b.encode(*pairs[1])[2]
    assert [t.data_ptr() for t in n2] != [t.data_ptr() for t in b._graph_state["net"]]
    for x, y in zip(net, n2):
        assert torch.equal(x, y)
    lo, up = a.iterate(f1, f2, net, inp, 6)
    assert torch.equal(up, b(*pairs[1], iters=6, test_mode=True)[1])


@torch.no_grad()
def test_paired_context_heads_equal_the_single_launches():
    """extractor.paired_heads (merged first layers, paired residual / output layers; core/extractor.py:246-266, :284-296) ==
    the heads run one after the other, for the three scales' head shapes."""
    import _synth
    from test_gpu_parity import G, _raft
    from dkt_stereo_amd import extractor as ex
    m, _ = _raft()
    i1, _ = _synth.image_pair(4, 2, 192, 320, 12)
    x = (2 * (G(i1) / 255.0) - 1.0).contiguous()
    keep = ex.PAIR_HEADS
    try:
        ex.PAIR_HEADS = True
        got = m.cnet(x, num_layers=3)
        ex.PAIR_HEADS = False
        want = m.cnet(x, num_layers=3)
    finally:
        ex.PAIR_HEADS = keep
    for g, w in zip(got, want):
        for u, v in zip(g, w):
            assert torch.equal(u, v)


# ---------------------------------------------------------------------------------------------------------------------------
# Weights-stationary 64 -> 64 convolution of the encoders' full-resolution layers (csrc/conv_ws.h; core/extractor.py:122-143,
# 46-60): against fp64 and against the streaming kernel it replaces (same split, another summation order), every epilogue the
# encoders use, partial tiles, several images per launch, launch-to-launch reproducibility.
# ---------------------------------------------------------------------------------------------------------------------------
def _ws_forms(B, H, W, seed):
    import torch.nn as nn
    from dkt_stereo_amd import conv
    from dkt_stereo_amd.extractor import instance_norm_params
    g = torch.Generator(device=DEV).manual_seed(seed)
    torch.manual_seed(seed)                 # the layer's initialisation
    layer = nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    x = torch.randn(B, 64, H, W, device=DEV, generator=g) * 3.0 + 0.5
    res = torch.randn(B, 64, H, W, device=DEV, generator=g).relu()
    p = instance_norm_params(nn.InstanceNorm2d(64), x)
    out = {}
    with torch.no_grad():
        out["relu"] = conv.conv2d(x, layer, relu=True)
        out["join"] = conv.conv2d_fused(x, layer, relu=True, residual=res)
        y, st = conv.conv2d_stats(x, layer, in_norm=p)
        out["in_norm"] = y
        out["params"] = instance_norm_params(nn.InstanceNorm2d(64), y, st)
        y, st = conv.conv2d_stats(x, layer)
        out["plain_stats"] = y
        out["plain_params"] = instance_norm_params(nn.InstanceNorm2d(64), y, st)
        out["in_norm_join"] = conv.conv2d_fused(x, layer, relu=False, residual=res, in_norm=p)
    return layer, x, res, out


@pytest.mark.parametrize("B,H,W", [(1, 264, 544),      # 561 tiles: every block two or three
                                   (3, 130, 333),      # partial tiles on both axes, three images (per-image norm parameters)
                                   (2, 736, 1248)])    # the benchmark's feature-encoder launch
def test_weights_stationary_conv_matches_fp64_and_the_streaming_kernel(B, H, W, monkeypatch):
    monkeypatch.setenv("DKT_CONV_WS", "1")
    layer, x, res, ws = _ws_forms(B, H, W, 7)
    monkeypatch.setenv("DKT_CONV_WS", "0")
    _, _, _, st = _ws_forms(B, H, W, 7)
    w, b = layer.weight.detach().double(), layer.bias.detach().double()
    ref = F.conv2d(x.double(), w, b, padding=1)
    xn = F.instance_norm(x.double()).relu()
    refn = F.conv2d(xn, w, b, padding=1)
    want = {"relu": ref.relu(), "join": (res.double() + ref.relu()).relu(), "in_norm": refn, "plain_stats": ref,
            "in_norm_join": (res.double() + refn).relu()}
    for k, r in want.items():
        assert _rel(ws[k], r) < 2.5e-6, k                   # the split-fp16 class (tests/test_gpu_conv.py's bound)
        assert _rel(ws[k], st[k].double()) < 2.5e-6, k       # the kernel it replaces
    for k in ("params", "plain_params"):                     # (mean, 1/std) from the epilogue's partial sums
        assert _rel(ws[k], st[k].double()) < 4e-6, k
    mean = refn.mean(dim=(2, 3)).reshape(-1)
    istd = (refn.var(dim=(2, 3), unbiased=False) + 1e-5).rsqrt().reshape(-1)
    assert _rel(ws["params"][:, 0], mean) < 4e-6 and _rel(ws["params"][:, 1], istd) < 4e-6


def test_weights_stationary_conv_is_reproducible_and_gated(monkeypatch):
    import torch.nn as nn
    from dkt_stereo_amd import conv
    monkeypatch.setenv("DKT_CONV_WS", "1")
    _, _, _, a = _ws_forms(2, 300, 500, 11)
    _, _, _, b = _ws_forms(2, 300, 500, 11)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # images below 192 tiles, other widths and strides stay on the streaming kernel: same bits with the switch on or off
    torch.manual_seed(3)
    for cin, cout, H, W, stride in ((64, 64, 96, 160, 1), (64, 96, 264, 544, 1), (64, 64, 264, 544, 2)):
        layer = nn.Conv2d(cin, cout, 3, padding=1, stride=stride).to(DEV)
        x = torch.randn(1, cin, H, W, device=DEV)
        with torch.no_grad():
            monkeypatch.setenv("DKT_CONV_WS", "1")
            y1 = conv.conv2d(x, layer, relu=True)
            monkeypatch.setenv("DKT_CONV_WS", "0")
            y0 = conv.conv2d(x, layer, relu=True)
        assert torch.equal(y0, y1), (cin, cout, H, W, stride)


@pytest.mark.parametrize("cout", [96, 72])
def test_ninety_six_channel_tile_matches_fp64_in_every_form(cout):
    """65 .. 96 output channels run on a 96-channel wave tile (conv2d.hip: launch_conv<3, 1, 4, 1, P, 3>): plain / ReLU /
    residual join / instance-norm input + output statistics against fp64 (the layers of core/extractor.py's layer2)."""
    import torch.nn as nn
    from dkt_stereo_amd import conv
    from dkt_stereo_amd.extractor import instance_norm_params
    torch.manual_seed(cout)
    B, H, W = 2, 93, 157
    layer = nn.Conv2d(96, cout, 3, padding=1).to(DEV)
    x = torch.randn(B, 96, H, W, device=DEV) * 2 + 0.3
    res = torch.randn(B, cout, H, W, device=DEV).relu()
    w, b = layer.weight.detach().double(), layer.bias.detach().double()
    with torch.no_grad():
        ref = F.conv2d(x.double(), w, b, padding=1)
        assert _rel(conv.conv2d(x, layer), ref) < 2.5e-6
        assert _rel(conv.conv2d(x, layer, relu=True), ref.relu()) < 2.5e-6
        assert _rel(conv.conv2d_fused(x, layer, relu=True, residual=res), (res.double() + ref.relu()).relu()) < 2.5e-6
        p = instance_norm_params(nn.InstanceNorm2d(96), x)
        refn = F.conv2d(F.instance_norm(x.double()).relu(), w, b, padding=1)
        y, st = conv.conv2d_stats(x, layer, in_norm=p)
        assert _rel(y, refn) < 2.5e-6
        got = instance_norm_params(nn.InstanceNorm2d(cout), y, st)
        mean = refn.mean(dim=(2, 3)).reshape(-1)
        istd = (refn.var(dim=(2, 3), unbiased=False) + 1e-5).rsqrt().reshape(-1)
        assert _rel(got[:, 0], mean) < 4e-6 and _rel(got[:, 1], istd) < 4e-6
