"""-m gpu: the round-3 path -- the C8S convolution (csrc/conv_c8.hip), its producers and the loop built on them.

Oracles: an fp64 convolution of the same operands (bound 3e-6 relative: the split-fp16 arithmetic's class, the round-2
kernel measures 1.6-1.9e-6), the round-2 kernels (same arithmetic in another accumulation order), torch's own
resampling operators (bit-exact: the kernels restate ATen's operation order), and the reference fixtures through the full
model (final disparity <= 1e-3, north_star).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases
import _synth
from test_gpu_parity import DEV, G, _raft, maxabs

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max())


def _c8():
    from dkt_stereo_amd import conv_c8
    return conv_c8


@torch.no_grad()
@pytest.mark.parametrize("shape", [(1, 40, 72), (2, 50, 70), (1, 184, 312)])
def test_c8s_pack_roundtrip_and_border(shape):
    """fp32 -> C8S -> fp32 keeps 22 significant bits; the border and the padding channels stay zero."""
    c8 = _c8()
    B, H, W = shape
    torch.manual_seed(0)
    x = torch.randn(B, 70, H, W, device=DEV) * torch.logspace(-3, 3, 70, device=DEV).view(1, 70, 1, 1)
    a = c8.pack(x)
    assert a.t.shape[1] == 2 * 5 and a.t.shape[3:] == (c8.c8_dims(H, W)[0], c8.c8_dims(H, W)[1], 8)
    y = c8.unpack(a)
    # 22 significant bits above 2^-3, an absolute resolution of 2^-25 below (conv.py's stated range of the split operands)
    assert float(((y - x).abs() / x.abs().clamp_min(2.0 ** -3)).max()) <= 2.0 ** -21
    t = a.t.clone()
    t[:, :, :, 1:H + 1, 1:W + 1, :] = 0
    assert float(t.abs().max()) == 0.0
    # channels 70..79 of the last group are padding
    assert float(a.t[:, 8:, :, :, :, 6:].abs().max()) == 0.0


@torch.no_grad()
@pytest.mark.parametrize("case", [
    (1, 64, 96, [128, 128, 128], 256), (1, 184, 312, [128, 128, 128], 256), (1, 184, 312, [128, 128, 128], 128),
    (1, 184, 312, [64, 64], 126), (1, 92, 156, [128, 128, 128], 256), (1, 46, 78, [128, 128], 256),
    (2, 50, 70, [64], 64), (1, 33, 37, [16, 48], 40), (3, 17, 100, [80], 200)])
def test_conv_c8_matches_fp64_for_every_tile_shape(case):
    """conv(cat(srcs)) on the C8S kernel, every tile configuration, fp32 NCHW and C8S destinations, vs an fp64 convolution."""
    c8 = _c8()
    B, H, W, chs, cout = case
    torch.manual_seed(1)
    xs = [torch.randn(B, c, H, W, device=DEV) for c in chs]
    layer = torch.nn.Conv2d(sum(chs), cout, 3, padding=1).to(DEV)
    ref = F.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), padding=1)
    acts = [c8.pack(x) for x in xs]
    for cfg in (0, 1, 2, 3, 4, 5, 6):
        for relu in (False, True):
            want = ref.clamp_min(0) if relu else ref
            y = c8.conv2d_c8(acts, layer, relu=relu, cfg=cfg)
            oc = c8.ActC8(B, cout, H, W, DEV)
            c8.conv2d_c8(acts, layer, relu=relu, out_c8=oc, cfg=cfg)
            e1, e2 = _rel(y, want), _rel(c8.unpack(oc), want)
            assert e1 <= 3e-6 and e2 <= 3e-6, (case, cfg, relu, e1, e2)
            t = oc.t.clone()
            t[:, :, :, 1:H + 1, 1:W + 1, :] = 0
            assert float(t.abs().max()) == 0.0, "C8S border written"


@torch.no_grad()
def test_conv_c8_gates_tail_and_pair_match_round2_kernels():
    """ConvGRU gate epilogues (NCHW and C4 operands), the motion encoder's cat tail and two problems per launch, against the
    round-2 kernels (same arithmetic, another accumulation order: <= 3e-6 relative)."""
    from dkt_stereo_amd import conv
    c8 = _c8()
    torch.manual_seed(2)
    B, H, W = 1, 96, 160
    h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
    x1, x2 = torch.randn(B, 128, H, W, device=DEV), torch.randn(B, 128, H, W, device=DEV)
    cz, cr, cq = (torch.randn(B, 128, H, W, device=DEV) for _ in range(3))
    zr = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
    ql = torch.nn.Conv2d(384, 128, 3, padding=1).to(DEV)
    z0, rh0 = conv.conv2d_gate_zr([h, x1, x2], zr, cz, cr, h)
    hn0 = conv.conv2d_gate_out([rh0, x1, x2], ql, cq, z0, h)
    ah, a1, a2 = c8.pack(h), c8.pack(x1), c8.pack(x2)
    for cfg_zr, cfg_q in ((1, 2), (2, 3), (4, 4), (6, 6)):
        rh_c8, rh = c8.ActC8(B, 128, H, W, DEV), torch.empty_like(h)
        z = c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=rh_c8, rh=rh, cfg=cfg_zr)
        hn, hn_c8 = torch.empty_like(h), c8.ActC8(B, 128, H, W, DEV)
        c8.gate_out([rh_c8, a1, a2], ql, cq, z, h, hn, out_c8=hn_c8, cfg=cfg_q)
        errs = (_rel(z, z0), _rel(rh, rh0), _rel(c8.unpack(rh_c8), rh0), _rel(hn, hn0), _rel(c8.unpack(hn_c8), hn0))
        assert max(errs) <= 3e-6, (cfg_zr, cfg_q, errs)
    # C4 operands
    h4, cz4, cr4, cq4 = (c8.to_c4(t) for t in (h, cz, cr, cq))
    rh_c8 = c8.ActC8(B, 128, H, W, DEV)
    z4 = c8.gate_zr([ah, a1, a2], zr, cz4, cr4, h4, rh_c8=rh_c8, cfg=1, f32_c4=True)
    hn4 = torch.empty_like(h4)
    c8.gate_out([rh_c8, a1, a2], ql, cq4, z4, h4, hn4, cfg=2, f32_c4=True)
    assert _rel(c8.from_c4(z4), z0) <= 3e-6 and _rel(c8.from_c4(hn4), hn0) <= 3e-6
    # in-place state update (out aliases h), as the loop uses it
    h2 = h.clone()
    c8.gate_out([rh_c8, a1, a2], ql, cq, z0, h2, h2, cfg=2)
    assert _rel(h2, hn0) <= 3e-6
    # tail = torch.cat([out, flow]) (core/update.py:85)
    enc = torch.nn.Conv2d(128, 126, 3, padding=1).to(DEV)
    c1, f1, flow = torch.randn(B, 64, H, W, device=DEV), torch.randn(B, 64, H, W, device=DEV), torch.randn(B, 2, H, W, device=DEV)
    mf = c8.ActC8(B, 128, H, W, DEV)
    c8.conv2d_c8([c8.pack(c1), c8.pack(f1)], enc, relu=True, out_c8=mf, tail=flow)
    want = torch.cat([conv.conv2d([c1, f1], enc, relu=True), flow], 1)
    assert _rel(c8.unpack(mf), want) <= 3e-6
    # two problems in one launch == two launches (bit-identical: same kernel, same tiles)
    xs, hs = torch.randn(1, 256, 24, 40, device=DEV), torch.tanh(torch.randn(1, 128, 24, 40, device=DEV))
    czs, crs = torch.randn(1, 128, 24, 40, device=DEV), torch.randn(1, 128, 24, 40, device=DEV)
    zr2 = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
    ahs, axs = c8.pack(hs), c8.pack(xs)
    za, zb = torch.empty_like(h), torch.empty_like(hs)
    ra, rb = c8.ActC8(B, 128, H, W, DEV), c8.ActC8(1, 128, 24, 40, DEV)
    da = c8.desc([ah, a1, a2], zr, out=za, epilogue=1, e0=cz, e1=cr, h=h, out2_c8=ra)
    db = c8.desc([ahs, axs], zr2, out=zb, epilogue=1, e0=czs, e1=crs, h=hs, out2_c8=rb)
    c8.launch_pair(da, db, za, 1)
    rb1 = c8.ActC8(1, 128, 24, 40, DEV)
    zb1 = c8.gate_zr([ahs, axs], zr2, czs, crs, hs, rh_c8=rb1, cfg=1)
    ra1 = c8.ActC8(B, 128, H, W, DEV)
    za1 = c8.gate_zr([ah, a1, a2], zr, cz, cr, h, rh_c8=ra1, cfg=1)
    assert torch.equal(za, za1) and torch.equal(zb, zb1) and torch.equal(ra.t, ra1.t) and torch.equal(rb.t, rb1.t)


@torch.no_grad()
def test_c8s_producers_match_their_fp32_twins():
    """pool2x / interp / 7x7 stem / lookup+convc1 writing C8S directly == the fp32 kernels followed by the split."""
    from dkt_stereo_amd import conv
    from dkt_stereo_amd.corr import CorrBlock1D
    from dkt_stereo_amd.update import interp, pool2x
    c8 = _c8()
    torch.manual_seed(3)
    B, H, W = 2, 46, 78
    x = torch.randn(B, 128, H, W, device=DEV)
    big = torch.randn(B, 128, 92, 156, device=DEV)
    p = c8.pool2x_c8(big, c8.ActC8(B, 128, H, W, DEV))
    assert torch.equal(p.t, c8.pack(pool2x(big)).t)
    assert torch.equal(pool2x(big), F.avg_pool2d(big, 3, stride=2, padding=1))
    u = c8.interp_c8(x, c8.ActC8(B, 128, 92, 156, DEV))
    assert torch.equal(u.t, c8.pack(interp(x, big)).t)
    stem = torch.nn.Conv2d(2, 64, 7, padding=3).to(DEV)
    flow = torch.randn(B, 2, 92, 156, device=DEV) * 5
    s = c8.stem7_c8(flow, stem, c8.ActC8(B, 64, 92, 156, DEV))
    assert torch.equal(s.t, c8.pack(conv.conv2d(flow, stem, relu=True)).t)
    f1, f2 = torch.randn(B, 256, 40, 96, device=DEV), torch.randn(B, 256, 40, 96, device=DEV)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    xs = torch.arange(96, device=DEV, dtype=torch.float32).view(1, 1, 1, 96).expand(B, 1, 40, 96)
    ys = torch.arange(40, device=DEV, dtype=torch.float32).view(1, 1, 40, 1).expand(B, 1, 40, 96)
    coords = torch.cat([xs - 30 * torch.rand(B, 1, 40, 96, device=DEV), ys], 1).contiguous()
    c1 = torch.nn.Conv2d(36, 64, 1).to(DEV)
    got = blk.lookup_conv1x1(coords, c1, out_c8=c8.ActC8(B, 64, 40, 96, DEV))
    assert got is not None and torch.equal(got.t, c8.pack(blk.lookup_conv1x1(coords, c1)).t)


@torch.no_grad()
def test_fused_flow_head_matches_two_layers():
    """FlowHead with conv1's ReLU output reduced against conv2's taps in conv1's epilogue (x output only) == conv2(relu(conv1(h)))
    on the exact kernels, including the coordinate update and flow = coords1 - coords0 of raft_stereo.py:165-168."""
    from dkt_stereo_amd import conv
    from dkt_stereo_amd.update import FlowHead, _leading_outputs
    c8 = _c8()
    torch.manual_seed(4)
    for (B, H, W) in ((1, 184, 312), (2, 45, 70)):
        fh = FlowHead(128, 256, 2).to(DEV)
        h = torch.tanh(torch.randn(B, 128, H, W, device=DEV))
        coords0 = torch.randn(B, 2, H, W, device=DEV) * 10
        want_c = coords0.clone() + 1.5
        want_d = conv.conv2d(conv.conv2d(h, fh.conv1, relu=True), _leading_outputs(fh.conv2, 1))
        want_c[:, :1] += want_d
        for cfg in (2,):
            coords1 = coords0.clone() + 1.5
            flow = torch.zeros(B, 2, H, W, device=DEV)
            c8.head([c8.pack(h)], fh.conv1, _leading_outputs(fh.conv2, 1), coords1[:, :1], diff=(coords0[:, :1], flow[:, :1]), cfg=cfg)
            assert float((coords1 - want_c).abs().max()) <= 2e-5 * max(1.0, float(want_d.abs().max()))
            assert torch.equal(flow[:, :1], coords1[:, :1] - coords0[:, :1])
            assert torch.equal(coords1[:, 1:], want_c[:, 1:])


@torch.no_grad()
def test_raft_c8_loop_matches_round2_loop_and_fixture(golden):
    """The whole forward at the benchmark workload on the C8S loop: within 1e-3 of the reference fixture, within 2e-4 of the
    round-2 loop, graph replay == first (eager + captured) forward, and a second pair through the same captured state."""
    c = _cases.E2E_CASES["736x1248_it32"]
    model, _ = _raft()
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"]))
    g = golden("raft_e2e")
    s = int(g["736x1248_it32/stride"])
    model.use_c8 = True
    _, up_first = model(i1, i2, iters=c["iters"], test_mode=True)
    _, up_replay = model(i1, i2, iters=c["iters"], test_mode=True)
    assert model._graph_state.get("c8") is not None, "the C8S loop did not run"
    assert torch.equal(up_first, up_replay)
    d_ref = maxabs(up_replay[:, :, ::s, ::s], g["736x1248_it32/flow_up"])
    model.use_c8 = False
    _, up_r2 = model(i1, i2, iters=c["iters"], test_mode=True)
    d_r2 = maxabs(up_replay, up_r2)
    print("C8S loop: vs reference fixture %.3e, vs round-2 loop %.3e" % (d_ref, d_r2))
    assert d_ref <= 1e-3 and d_r2 <= 2e-4
    # another pair, then the first one again, through the same captured graph
    model.use_c8 = True
    j1, j2 = (G(t) for t in _synth.image_pair(11, 1, c["H"], c["W"], 20))
    model(j1, j2, iters=c["iters"], test_mode=True)
    _, up_again = model(i1, i2, iters=c["iters"], test_mode=True)
    assert torch.equal(up_again, up_replay)


@torch.no_grad()
@pytest.mark.parametrize("shape", [(1, 8, 8, 5, 48), (2, 3, 12, 7, 100), (1, 5, 16, 3, 68), (1, 2, 4, 9, 200), (1, 40, 8, 6, 240)])
def test_gwc_volume_mfma_matches_oracle(shape, c_oracle):
    """ns-1: the group-wise correlation as a banded product on v_mfma_f32_16x16x4_f32 (gwc_mfma.hip) vs the C oracle
    (<= 2e-6 x scale: an fma chain against the oracle's sum of rounded products), ragged widths, batches, and inside the
    64-channel GwcNet buffer (a batch stride larger than the volume)."""
    from dkt_stereo_amd import _ffi, submodule as sm
    B, G_, cpg, H, W = shape
    a = _synth.normal((B, G_ * cpg, H, W), 5, "a")
    b = _synth.normal((B, G_ * cpg, H, W), 5, "b")
    want = c_oracle.gwc_volume(a, b, 48, G_)
    ga, gb = G(a), G(b)
    vol = torch.full((B, G_ + 3, 48, H, W), 7.0, device=DEV)
    rc = _ffi.lib().dkt_gwc_volume_mfma(ga.data_ptr(), gb.data_ptr(), vol.data_ptr(), B, G_ * cpg, H, W, 48, G_, vol.stride(0),
                                        _ffi.device_of(vol), _ffi.stream_of(vol))
    assert rc == 0
    got = vol[:, :G_].cpu().numpy()
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    assert float((vol[:, G_:] - 7.0).abs().max()) == 0.0            # nothing written behind the G planes
    with sm.gwc_mode("mfma"):
        assert torch.equal(sm.build_gwc_volume(ga, gb, 48, G_), vol[:, :G_].contiguous())
    # shapes outside the MFMA form are refused (the Python wrapper then takes the VALU kernel)
    v2 = torch.empty((B, G_, 40, H, W), device=DEV)
    assert _ffi.lib().dkt_gwc_volume_mfma(ga.data_ptr(), gb.data_ptr(), v2.data_ptr(), B, G_ * cpg, H, W, 40, G_, v2.stride(0),
                                          _ffi.device_of(v2), _ffi.stream_of(v2)) == -7


@torch.no_grad()
@pytest.mark.parametrize("shape", [(2, 64, 40, 72), (1, 20, 33, 37)])
def test_instance_norm_join_c8_matches_torch(shape):
    """The one-pass instance-norm glue (normalise [+ReLU] [+ residual join]) -> fp32 and C8S, against torch's instance_norm."""
    from dkt_stereo_amd import extractor as ex
    c8 = _c8()
    B, C, H, W = shape
    torch.manual_seed(3)
    norm = torch.nn.InstanceNorm2d(C)
    c = torch.randn(B, C, H, W, device=DEV) * 3 + 1
    a = torch.randn(B, C, H, W, device=DEV) * 2 - 0.5
    pc, pa = ex.instance_norm_params(norm, c), ex.instance_norm_params(norm, a)
    nc, na = F.instance_norm(c.double()), F.instance_norm(a.double())
    for want, kw in ((nc.clamp_min(0), dict(c_relu=True)),
                     (nc, dict(c_relu=False)),
                     ((a.double() + nc.clamp_min(0)).clamp_min(0), dict(c_relu=True, a=a)),
                     ((na.clamp_min(0) + nc.clamp_min(0)).clamp_min(0), dict(c_relu=True, a=a, a_params=pa, a_relu=True)),
                     ((na + nc.clamp_min(0)).clamp_min(0), dict(c_relu=True, a=a, a_params=pa, a_relu=False))):
        y = torch.empty_like(c)
        d = c8.ActC8(B, C, H, W, DEV)
        c8.norm_join_c8(c, pc, y=y, dst=d, **kw)
        assert _rel(y, want) <= 2e-6, kw.keys()
        # the C8S twin is the split of exactly those fp32 values
        assert torch.equal(c8.unpack(d), c8.unpack(c8.pack(y)))
        t = d.t.clone()
        t[:, :, :, 1:H + 1, 1:W + 1, :] = 0
        assert float(t.abs().max()) == 0.0


@torch.no_grad()
@pytest.mark.parametrize("case", [(1, 64, 96, 64, 64), (2, 50, 70, 64, 96), (1, 33, 37, 48, 40)])
def test_conv_c8_residual_epilogue_matches_fp64(case):
    """Epilogue 4: relu(res + relu(conv + bias)) -> fp32 (in place over res) and C8S, every tile shape."""
    c8 = _c8()
    B, H, W, cin, cout = case
    torch.manual_seed(4)
    x = torch.randn(B, cin, H, W, device=DEV)
    res = torch.randn(B, cout, H, W, device=DEV)
    layer = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
    want = (res.double() + F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=1).clamp_min(0)).clamp_min(0)
    a = c8.pack(x)
    for cfg in (0, 1, 2, 3, 4, 5, 6):
        r = res.clone()
        oc = c8.ActC8(B, cout, H, W, DEV)
        y = c8.residual_c8([a], layer, r, relu=True, out=r, out_c8=oc, cfg=cfg)
        assert y is r and _rel(y, want) <= 3e-6, (case, cfg, _rel(y, want))
        assert torch.equal(c8.unpack(oc), c8.unpack(c8.pack(y))), (case, cfg)
        assert float(oc.t[:, :, :, 0].abs().max()) == 0.0 and float(oc.t[:, :, :, :, 0].abs().max()) == 0.0


@torch.no_grad()
def test_encoder_layer1_on_c8_matches_round2_path(monkeypatch):
    """conv1 + layer1 of both encoders on conv_c8 (extractor._layer1_c8, DKT_C8_ENCODER=1) against the round-2 kernels:
    the same split-fp16 arithmetic in another accumulation order."""
    from dkt_stereo_amd import extractor as ex
    m, _ = _raft(DEV)
    torch.manual_seed(5)
    x = torch.rand(2, 3, 96, 160, device=DEV) * 2 - 1
    monkeypatch.setattr(ex, "C8_ENCODER", False)
    f0, c0 = m.fnet._trunk(x), m.cnet._trunk(x[:1])
    monkeypatch.setattr(ex, "C8_ENCODER", True)
    monkeypatch.setattr(ex, "C8_ENCODER_MIN_PIXELS", 0)
    assert m.fnet._layer1_c8_kind(x) == 'instance' and m.cnet._layer1_c8_kind(x[:1]) == 'batch'
    for cfg in (3, 4):
        monkeypatch.setattr(ex, "C8_ENCODER_CFG", cfg)
        f1, c1 = m.fnet._trunk(x), m.cnet._trunk(x[:1])
        assert _rel(f1, f0) <= 2e-5 and _rel(c1, c0) <= 2e-5, (cfg, _rel(f1, f0), _rel(c1, c0))
    # the cached C8S buffers are reused by a second call and follow another shape
    assert torch.equal(m.fnet._trunk(x), f1)
    x2 = torch.rand(1, 3, 64, 96, device=DEV) * 2 - 1
    monkeypatch.setattr(ex, "C8_ENCODER", False)
    g0 = m.fnet._trunk(x2)
    monkeypatch.setattr(ex, "C8_ENCODER", True)
    assert _rel(m.fnet._trunk(x2), g0) <= 2e-5


@torch.no_grad()
def test_igev_c8_loop_matches_round2_loop_and_fixture(golden, monkeypatch):
    """IGEV's refinement loop on the C8S kernels (loop_c8.C8LoopIGEV) at the cfg3 shapes, 32 iterations: against the round-2
    loop (same arithmetic, another accumulation order), the reference fixture (<= 1e-3), and itself replayed with a new
    volume of the same shapes."""
    from test_gpu_round2 import _igev_setup
    from dkt_stereo_amd import igev_loop, loop_c8
    c = _cases.IGEV_LOOP_CASES["kitti"]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    assert loop_c8.eligible_igev(blk, net[0].shape)
    g = golden("igev_loop")
    st = int(g["kitti/mask_stride"])
    monkeypatch.setattr(igev_loop, "USE_C8", False)
    r2_d, r2_m, r2_n = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, c["iters"], cache={})
    monkeypatch.setattr(igev_loop, "USE_C8", True)
    cache = {}
    d, m, n = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, c["iters"], cache=cache)
    assert getattr(cache["state"], "c8", None) is not None
    e_d, e_m, e_n = maxabs(d, r2_d), maxabs(m, r2_m), max(maxabs(a, b) for a, b in zip(n, r2_n))
    print("igev c8 loop vs round-2 loop: disp %.2e mask %.2e net %.2e; vs fixture %.2e" % (e_d, e_m, e_n, maxabs(d, g["kitti/disp"])))
    assert e_d <= 2e-4 and e_m <= 2e-4 and e_n <= 2e-4
    assert maxabs(d, g["kitti/disp"]) <= 1e-3 and maxabs(m[:, :, ::st, ::st], g["kitti/mask"]) <= 1e-3
    d2, m2, _ = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, c["iters"], cache=cache)
    assert torch.equal(d2, d) and torch.equal(m2, m)            # second call: all units replayed from the captured graph


def test_few_output_kernel_is_exact_beside_lds_holders_without_the_exclusive_claim():
    """DESIGN 3.4 / ADVICE r02: tools/stress_lds_dma.py with DKT_FEW_LDS_EXACT=1 -- the few-output head kernel requests only
    the LDS it needs, so blocks of the 1/16-resolution GRU convolution on a second stream share its CUs (the situation that
    gave 1000 wrong launches of 1200 with the packed-FMA build).  The scalar-FMA build (-fno-slp-vectorize) must be exact in
    every launch, and so must the co-resident convolution."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DKT_FEW_LDS_EXACT="1")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_lds_dma.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("exact-LDS=1")]
    assert len(lines) == 2, res.stdout[-2000:]
    for ln in lines:
        assert "few-kernel results wrong 0/1200" in ln and "co-resident conv results wrong 0/600" in ln, ln


@torch.no_grad()
@pytest.mark.parametrize("shape", [(1, 40, 72, 48), (2, 23, 100, 32), (1, 184, 312, 48)])
def test_geo_lookup_fused_with_convc1(shape):
    """IGEV's geometry lookup fused with the motion encoder's 1x1 layer (dkt_geo_lookup_conv1x1): the tap output is the plain
    lookup bit for bit, the 64 channels match an fp64 evaluation of the 1x1 layer on it, the C8S output is the split of the
    fp32 one; ragged widths, batch 2, smooth and random disparities."""
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    c8 = _c8()
    B, H, W, D = shape
    m1, m2 = G(_synth.normal((B, 24, H, W), 11, "m1")), G(_synth.normal((B, 24, H, W), 11, "m2"))
    geo = G(_synth.normal((B, 8, D, H, W), 12, "geo"))
    fn = Combined_Geo_Encoding_Volume(m1, m2, geo, radius=4, num_levels=2)
    coords = torch.arange(W, device=DEV).float().view(1, 1, W, 1).repeat(B, H, 1, 1)
    layer = torch.nn.Conv2d(162, 64, 1).to(DEV)
    torch.manual_seed(6)
    xs = torch.arange(W, device=DEV).float().view(1, 1, 1, W)
    for kind in ("smooth", "random", "edge"):
        if kind == "smooth":
            disp = (3.0 + (D - 8.0) * xs / W + 0.3 * torch.rand(B, 1, H, W, device=DEV)).contiguous()
        elif kind == "random":
            disp = torch.rand(B, 1, H, W, device=DEV) * (D + 10) - 5
        else:
            disp = torch.full((B, 1, H, W), -3.5, device=DEV)
            disp[:, :, ::2] = D + 2.25
        want_tap = fn(disp, coords)
        out, tap = fn.lookup_conv1x1(disp, coords, layer, relu=True, tap=True)
        assert torch.equal(tap, want_tap), kind
        ref = F.conv2d(want_tap.double(), layer.weight.double(), layer.bias.double()).clamp_min(0)
        assert _rel(out, ref) <= 2e-6, (kind, _rel(out, ref))
        dst = c8.ActC8(B, 64, H, W, DEV)
        assert fn.lookup_conv1x1(disp, coords, layer, relu=True, out_c8=dst) is dst
        assert torch.equal(c8.unpack(dst), c8.unpack(c8.pack(out))), kind
        t = dst.t.clone()
        t[:, :, :, 1:H + 1, 1:W + 1, :] = 0
        assert float(t.abs().max()) == 0.0
        # a disparity view with a batch stride (the loop keeps it in the tail of a wider buffer)
        wide = torch.zeros(B, 3, H, W, device=DEV)
        wide[:, 2:] = disp
        assert torch.equal(fn.lookup_conv1x1(wide[:, 2:], coords, layer, relu=True), out)
    # configurations outside the fused form are declined (the caller runs the two steps)
    assert fn.lookup_conv1x1(disp, coords, torch.nn.Conv2d(162, 96, 1).to(DEV)) is None


@pytest.mark.parametrize("case", [(2, 24, 40, 3, 48, 40), (1, 33, 37, 1, 36, 64), (1, 20, 28, 7, 2, 64), (1, 40, 72, 3, 384, 256)])
def test_conv2d_autograd_matches_torch(case):
    """conv.conv2d_autograd (forward and input gradient on this library's kernels, weight gradient on the vendor library)
    against torch's autograd of F.conv2d in fp64: value, d/dx, d/dw, d/db; with and without the fused ReLU."""
    from dkt_stereo_amd import conv
    B, H, W, k, cin, cout = case
    torch.manual_seed(7)
    layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(DEV)
    x0 = torch.randn(B, cin, H, W, device=DEV)
    gy = torch.randn(B, cout, H, W, device=DEV)
    for relu in (False, True):
        x = x0.clone().requires_grad_(True)
        y = conv.conv2d_autograd(x, layer, relu=relu)
        gx, gw, gb = torch.autograd.grad(y, [x, layer.weight, layer.bias], gy)
        xd = x0.double().requires_grad_(True)
        wd, bd = layer.weight.detach().double().requires_grad_(True), layer.bias.detach().double().requires_grad_(True)
        yd = F.conv2d(xd, wd, bd, padding=k // 2)
        yd = F.relu(yd) if relu else yd
        gxd, gwd, gbd = torch.autograd.grad(yd, [xd, wd, bd], gy.double())
        assert _rel(y, yd) <= 3e-6 and _rel(gx, gxd) <= 5e-6 and _rel(gw, gwd) <= 2e-5 and _rel(gb, gbd) <= 2e-5, \
            (case, relu, _rel(y, yd), _rel(gx, gxd), _rel(gw, gwd), _rel(gb, gbd))


@pytest.mark.parametrize("igev", [False, True])
def test_update_block_autograd_matches_oracle(igev):
    """Training through the update operator (tools/ft_dkt.py:223-242): one update-block call with autograd enabled --
    values and the gradients of a scalar loss with respect to the hidden states, the correlation features and a sample of
    the parameters -- against the CPU oracle's autograd (the oracle is pinned on the reference's modules)."""
    from types import SimpleNamespace
    from oracle import torch_oracle as to
    from dkt_stereo_amd.update import BasicMultiUpdateBlock, BasicMultiUpdateBlockIGEV
    cfg = dict(corr_levels=2 if igev else 4, corr_radius=4, n_downsample=2, n_gru_layers=3, hidden_dims=[128, 128, 128],
               slow_fast_gru=False)
    cls = BasicMultiUpdateBlockIGEV if igev else BasicMultiUpdateBlock
    blk = cls(SimpleNamespace(**cfg), hidden_dims=cfg["hidden_dims"])
    sd = _synth.torch_state_dict(_synth.shapes_of(blk), 21)
    blk.load_state_dict(sd)
    blk.to(DEV)
    H, W = 16, 24
    torch.manual_seed(8)
    net0 = [torch.tanh(torch.randn(1, 128, H >> i, W >> i)) for i in range(3)]
    inp = [[0.5 * torch.randn(1, 128, H >> i, W >> i) for _ in range(3)] for i in range(3)]
    corr0 = torch.randn(1, 162 if igev else 36, H, W)
    aux = torch.randn(1, 1 if igev else 2, H, W)
    wts = [torch.randn(1, 128, H >> i, W >> i) for i in range(3)]
    wd = torch.randn(1, 1 if igev else 2, H, W)
    wm = torch.randn(1, 32 if igev else 144, H, W)
    names = ["encoder.convc1.weight", "encoder.conv.bias", ("gru04" if igev else "gru08") + ".convz.weight",
             ("gru08" if igev else "gru16") + ".convq.weight", ("gru16" if igev else "gru32") + ".convr.bias",
             ("disp_head" if igev else "flow_head") + ".conv2.weight", ("mask_feat_4.0" if igev else "mask.0") + ".weight"]

    def loss_of(net, mask, delta, to_dev):
        t = lambda a: a.to(to_dev)
        return sum((n * t(w)).sum() for n, w in zip(net, wts)) + (delta * t(wd)).sum() + (mask * t(wm)).sum()

    # ---- ours (GPU)
    net_g = [t.to(DEV).requires_grad_(True) for t in net0]
    corr_g = corr0.to(DEV).requires_grad_(True)
    kw = dict(disp=aux.to(DEV)) if igev else dict(flow=aux.to(DEV))
    net, mask, delta = blk(list(net_g), [[t.to(DEV) for t in s] for s in inp], corr_g, **kw)
    params = dict(blk.named_parameters())
    got = torch.autograd.grad(loss_of(net, mask, delta, DEV), net_g + [corr_g] + [params[n] for n in names])
    # ---- oracle (CPU, fp64)
    sdd = {("ub." + k): v.double().requires_grad_(True) for k, v in sd.items()}
    net_c = [t.double().requires_grad_(True) for t in net0]
    corr_c = corr0.double().requires_grad_(True)
    o_net, o_mask, o_delta = to.update_block(sdd, "ub", 3, list(net_c), [[t.double() for t in s] for s in inp], corr_c,
                                             aux.double(), igev=igev)
    want = torch.autograd.grad(loss_of(o_net, o_mask, o_delta, "cpu"), net_c + [corr_c] + [sdd["ub." + n] for n in names])
    for a, b in zip(list(net) + [mask, delta], list(o_net) + [o_mask, o_delta]):
        assert _rel(a.detach().cpu(), b.detach()) <= 1e-5
    for name, a, b in zip(["net0", "net1", "net2", "corr"] + names, got, want):
        assert _rel(a.cpu(), b) <= 5e-5, (name, _rel(a.cpu(), b))


def test_raft_training_forward_matches_inference_and_oracle_gradients():
    """RAFTStereo.forward(test_mode=False) under autograd (tools/ft_dkt.py:223): the last of the per-iteration predictions
    equals the test_mode result; after ONE iteration (no detach inside the chain) the gradients of a scalar loss with respect
    to parameters of the update block, the context convolutions and the last layer of either encoder match the CPU oracle's
    autograd.  (The encoders' other layers are frozen here: trainable ones fall back to torch layer by layer, which works
    but spends minutes in the vendor library's kernel search on a fresh box.)"""
    from oracle import torch_oracle as to
    from dkt_stereo_amd.raft_stereo import BASE_CONFIG
    model, sd = _raft(DEV)
    names = ["update_block.gru08.convz.weight", "update_block.gru16.convq.bias", "update_block.encoder.convc1.bias",
             "update_block.flow_head.conv2.weight", "update_block.mask.2.weight", "context_zqr_convs.0.weight",
             "fnet.conv2.weight", "cnet.outputs08.0.1.weight"]
    for n, p in model.named_parameters():
        if (n.startswith("fnet.") or n.startswith("cnet.")) and n not in names:
            p.requires_grad_(False)
    i1, i2 = _synth.image_pair(5, 1, 64, 128, 12)
    g1, g2 = G(i1), G(i2)
    with torch.no_grad():
        _, want_up = model(g1, g2, iters=3, test_mode=True)
    out = model(g1, g2, iters=3, test_mode=False)
    assert isinstance(out, dict) and list(out) == ["disp_preds"]          # the reference's convention (raft_stereo.py:185-187)
    preds = out["disp_preds"]
    assert len(preds) == 3 and preds[-1].requires_grad
    assert maxabs(preds[-1].detach(), want_up) <= 1e-4
    wl = torch.randn(1, 1, 64, 128, generator=torch.Generator().manual_seed(9))
    params = dict(model.named_parameters())
    pred = model(g1, g2, iters=1, test_mode=False)["disp_preds"][0]
    got = torch.autograd.grad((pred * wl.to(DEV)).sum(), [params[n] for n in names])
    sdd = {k: v.clone() for k, v in sd.items()}
    for n in names:
        sdd[n].requires_grad_(True)
    fm1, fm2, net, inp = to.raft_prepare(sdd, dict(BASE_CONFIG), torch.from_numpy(i1), torch.from_numpy(i2))
    _, o_up = to.raft_iterations(sdd, dict(BASE_CONFIG), fm1, fm2, net, inp, 1)
    want = torch.autograd.grad((o_up * wl).sum(), [sdd[n] for n in names])
    assert maxabs(pred.detach(), o_up.detach()) <= 1e-4
    for n, a, b in zip(names, got, want):
        assert _rel(a.cpu(), b) <= 5e-4, (n, _rel(a.cpu(), b))
    with torch.no_grad():        # the reference's call convention without autograd: the same list, on the inference kernels
        plain = model(g1, g2, iters=3, test_mode=False)["disp_preds"]
    assert len(plain) == 3 and not plain[-1].requires_grad and maxabs(plain[-1], want_up) <= 1e-4


@torch.no_grad()
def test_normalize_pair_is_the_reference_arithmetic():
    """dkt_normalize_pair: 2 * (x / 255) - 1 for both images into the feature encoder's concatenated batch, bit for bit the
    reference's CPU arithmetic (raft_stereo.py:91-92); also from batch-strided views."""
    from dkt_stereo_amd.raft_stereo import RAFTStereo
    torch.manual_seed(10)
    a, b = torch.rand(2, 3, 40, 72, device=DEV) * 255, torch.rand(2, 3, 40, 72, device=DEV) * 255
    n1, n2, both = RAFTStereo._normalized_pair(a, b)
    assert both is not None and both.shape == (4, 3, 40, 72)
    # the reference arithmetic is the CPU's (true division; torch's GPU kernel multiplies by a rounded 1/255 instead)
    ref = lambda t: 2 * (t.cpu() / 255.0) - 1.0
    assert torch.equal(n1.cpu(), ref(a)) and torch.equal(n2.cpu(), ref(b))
    assert torch.equal(both, torch.cat([n1, n2], 0)) and n1.data_ptr() == both.data_ptr()
    wide = torch.rand(2, 2, 3, 40, 72, device=DEV) * 255
    m1, m2, _ = RAFTStereo._normalized_pair(wide[:, 0], wide[:, 1])
    assert torch.equal(m1.cpu(), ref(wide[:, 0])) and torch.equal(m2.cpu(), ref(wide[:, 1]))
