"""-m gpu: the HIP path (through the C ABI of libdktstereo.so) against
  (1) the committed reference outputs (tests/golden/*.npz),
  (2) the oracles on the same seeded inputs, and
  (3) size-independent properties at BASELINE.json's full sizes.

Tolerances (fp32 everywhere):
  * sampler-only kernels fed identical pyramids: BIT EXACT (same arithmetic);
  * anything behind a contraction over channels (corr build, convolutions):
    fp32 round-off, 4e-6 * scale (summation order differs from BLAS/MKLDNN);
  * final disparity maps: <= 1e-3 max-abs, the bound north_star states.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import _cases
import _synth
from oracle import torch_oracle as to

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def maxabs(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def flat(p):
    return p.reshape(p.shape[0], -1)


def test_native_library_is_what_runs():
    from dkt_stereo_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH)
    assert _ffi.lib().dkt_version() == 1
    maps = open("/proc/self/maps").read()
    assert "libdktstereo.so" in maps


# ---------------------------------------------------------------------------------
# RAFT correlation
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
@torch.no_grad()
def test_corr_build_and_lookup(name, golden, c_oracle):
    from dkt_stereo_amd.corr import CorrBlock1D, CorrBlockFast1D
    c = _cases.CORR_CASES[name]
    g = golden("corr")
    f1, f2, coords = _cases.corr_inputs(c)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    s = int(g[name + "/pyr_stride"])
    scale = max(1.0, float(np.abs(g[name + "/pyr0"]).max()))
    assert len(blk.corr_pyramid) == c["L"]
    for i in range(c["L"]):
        mine = flat(blk.corr_pyramid[i].cpu().numpy())
        assert blk.corr_pyramid[i].shape == (c["B"] * c["H"] * c["W"], 1, 1, c["W2"] >> i)
        assert maxabs(mine[::s], g["%s/pyr%d" % (name, i)]) <= 4e-6 * scale
    # pooled levels are exact functions of level 0
    pooled = c_oracle.pool_pyramid(flat(blk.corr_pyramid[0].cpu().numpy()), c["L"])
    for i in range(c["L"]):
        assert np.array_equal(pooled[i], flat(blk.corr_pyramid[i].cpu().numpy()))
    out = blk(G(coords))
    assert out.shape == (c["B"], c["L"] * (2 * c["r"] + 1), c["H"], c["W"]) and out.is_contiguous()
    assert maxabs(out, g[name + "/lookup"]) <= 1e-5 * scale
    # sampler arithmetic alone: HIP lookup on its own pyramid == C oracle on that pyramid, bit for bit
    mine_pyr = [flat(p.cpu().numpy()) for p in blk.corr_pyramid]
    assert np.array_equal(out.cpu().numpy(), c_oracle.corr1d_lookup(mine_pyr, coords, c["r"]))
    # static corr() and the reg_cuda flavour
    vol = CorrBlock1D.corr(G(f1), G(f2))
    assert vol.shape == (c["B"], c["H"], c["W"], 1, c["W2"])
    assert torch.equal(vol.reshape(-1, c["W2"]), blk.corr_pyramid[0].reshape(-1, c["W2"]))
    fast = CorrBlockFast1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    assert fast.corr_pyramid[0].shape == (c["B"], c["H"], c["W"], 1, c["W2"])
    assert torch.equal(fast(G(coords)), out)


@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
@torch.no_grad()
def test_corr_alt_and_cosine(name, golden):
    from dkt_stereo_amd.corr import CorrBlock1D_Cosine, PytorchAlternateCorrBlock1D
    c = _cases.CORR_CASES[name]
    g = golden("corr")
    f1, f2, coords = _cases.corr_inputs(c)
    scale = max(1.0, float(np.abs(g[name + "/pyr0"]).max()))
    alt = PytorchAlternateCorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    assert maxabs(alt(G(coords)), g[name + "/alt"]) <= 1e-5 * scale
    cos = CorrBlock1D_Cosine(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    s = int(g[name + "/pyr_stride"])
    assert maxabs(flat(cos.corr_pyramid[0].cpu().numpy())[::s], g[name + "/cos0"]) <= 2e-6
    assert maxabs(cos(G(coords)), g[name + "/coslookup"]) <= 4e-6


@torch.no_grad()
def test_lookup_strided_coords_and_errors():
    """coords is usually a (B,2,H,W) tensor whose channel 0 is read in place."""
    from dkt_stereo_amd import _ffi
    from dkt_stereo_amd.corr import CorrBlock1D
    c = _cases.CORR_CASES["small"]
    f1, f2, coords = _cases.corr_inputs(c)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    a = blk(G(coords))
    wide = torch.zeros(c["B"], 5, c["H"], c["W"], device=DEV)
    wide[:, :2] = G(coords)
    assert torch.equal(blk(wide[:, :2]), a)                 # batch stride 5*H*W
    assert torch.equal(blk(G(coords).permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)), a)  # non-dense
    with pytest.raises(_ffi.DktError):
        CorrBlock1D(G(f1), G(f2), num_levels=9, radius=4)
    with pytest.raises(_ffi.DktError):
        CorrBlock1D(G(f1), G(f2), num_levels=2, radius=9)(G(coords))
    with torch.enable_grad():       # the inference-only variants refuse autograd inputs, they never detach silently
        from dkt_stereo_amd.corr import CorrBlock1D_Cosine
        with pytest.raises(_ffi.DktError):
            CorrBlock1D_Cosine(G(f1).requires_grad_(), G(f2), num_levels=2, radius=4)
        blk_g = CorrBlock1D(G(f1).requires_grad_(), G(f2), num_levels=2, radius=4)     # differentiable (8f-2)
        assert blk_g.corr_pyramid[0].requires_grad


# ---------------------------------------------------------------------------------
# IGEV geometry volume
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.GEO_CASES))
@torch.no_grad()
def test_geo_lookup(name, golden, c_oracle):
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    c = _cases.GEO_CASES[name]
    g = golden("geo")
    m1, m2, geo, disp, coords = _cases.geo_inputs(c)
    fn = Combined_Geo_Encoding_Volume(G(m1), G(m2), G(geo), num_levels=c["L"], radius=c["r"])
    want = g[name + "/lookup"]
    scale = max(1.0, float(np.abs(want).max()))
    assert maxabs(flat(fn.init_corr_pyramid[0].cpu().numpy()), g[name + "/init0"]) <= 4e-6 * scale
    out = fn(G(disp), G(coords))
    assert out.shape == want.shape
    assert maxabs(out, want) <= 1e-5 * scale
    # bit-exact against the C oracle when both read the HIP-built pyramids
    gp = []
    for p in fn.geo_volume_pyramid:
        b, ch, d, h, w = p.shape
        gp.append(p.permute(0, 3, 4, 1, 2).reshape(b * h * w * ch, d).cpu().numpy())
    ip = [flat(p.cpu().numpy()) for p in fn.init_corr_pyramid]
    assert np.array_equal(out.cpu().numpy(), c_oracle.geo_lookup(gp, ip, disp, coords, c["C"], c["r"]))


# ---------------------------------------------------------------------------------
# cost volumes
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.GWC_CASES))
@torch.no_grad()
def test_gwc_volume(name, golden, c_oracle):
    from dkt_stereo_amd.submodule import build_gwc_volume
    c = _cases.GWC_CASES[name]
    a, b = _cases.volume_inputs(c)
    from dkt_stereo_amd import submodule as sm
    vol = build_gwc_volume(G(a), G(b), c["D"], c["G"])          # default: the MFMA form where it applies (D = 48, C/G % 4 == 0)
    assert maxabs(vol, golden("volumes")["gwc/" + name]) <= 2e-6
    with sm.gwc_mode("exact"):
        vol = build_gwc_volume(G(a), G(b), c["D"], c["G"])
    assert maxabs(vol, golden("volumes")["gwc/" + name]) <= 2e-6
    assert np.array_equal(vol.cpu().numpy(), c_oracle.gwc_volume(a, b, c["D"], c["G"]))  # same order -> bit exact


@pytest.mark.parametrize("name", list(_cases.CONCAT_CASES))
@torch.no_grad()
def test_concat_volume(name, golden):
    from dkt_stereo_amd.submodule import build_concat_volume, build_concat_volume_igev
    c = _cases.CONCAT_CASES[name]
    a, b = _cases.volume_inputs(c)
    g = golden("volumes")
    assert np.array_equal(build_concat_volume(G(a), G(b), c["D"]).cpu().numpy(), g["concat_gwcnet/" + name])
    assert np.array_equal(build_concat_volume_igev(G(a), G(b), c["D"]).cpu().numpy(), g["concat_igev/" + name])


@torch.no_grad()
def test_fused_gwc_concat_buffer(golden):
    from dkt_stereo_amd.submodule import build_gwc_concat_volume
    cg, cc = _cases.GWC_CASES["igev"], dict(_cases.CONCAT_CASES["gc"])
    a, b = _cases.volume_inputs(cg)
    ca, cb = _cases.volume_inputs(cc)
    vol = build_gwc_concat_volume(G(a), G(b), G(ca), G(cb), cg["D"], cg["G"])
    g = golden("volumes")
    want = np.concatenate([g["gwc/igev"], g["concat_gwcnet/gc"]], axis=1)
    assert vol.shape == want.shape
    assert maxabs(vol, want) <= 2e-6


# ---------------------------------------------------------------------------------
# update operator
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.GRU_CASES))
@torch.no_grad()
def test_conv_gru(name, golden):
    from dkt_stereo_amd.update import ConvGRU
    c = _cases.GRU_CASES[name]
    h, czrq, xs = _cases.gru_inputs(c)
    hd = c["hidden"]
    gru = ConvGRU(hd, sum(c["inputs"]))
    sd = _synth.torch_state_dict({"g." + k: tuple(v.shape) for k, v in gru.state_dict().items()}, c["seed"])
    gru.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    gru.to(DEV).eval()
    cz, cr, cq = G(czrq).split(hd, dim=1)       # strided views, like the reference's context split
    out = gru(G(h), cz, cr, cq, *[G(x) for x in xs])
    assert maxabs(out, golden("gru")[name + "/h"]) <= 5e-6
    # reloading weights must invalidate the merged z|r cache
    sd2 = _synth.torch_state_dict({"g." + k: tuple(v.shape) for k, v in gru.state_dict().items()}, c["seed"] + 1)
    gru.load_state_dict({k[2:]: v.to(DEV) for k, v in sd2.items()}, strict=True)
    out2 = gru(G(h), cz, cr, cq, *[G(x) for x in xs])
    ref2 = to.conv_gru(sd2, "g", T(h), *T(czrq).split(hd, dim=1), *[T(x) for x in xs])
    assert maxabs(out2, ref2) <= 5e-6


@torch.no_grad()
def test_gate_kernels_vs_c_oracle(c_oracle):
    """dkt_gru_gate_zr/_out alone, vector and scalar (unaligned) paths, strided operands."""
    from dkt_stereo_amd import _ffi
    L = _ffi.lib()
    for B, Ch, H, W in ((2, 16, 6, 10), (1, 3, 5, 7)):
        HW = H * W
        az, ar, cz, cr = (_synth.normal((B, Ch, H, W), 5, n, scale=2.0) for n in ("az", "ar", "cz", "cr"))
        aq, cq = _synth.normal((B, Ch, H, W), 5, "aq", scale=2.0), _synth.normal((B, Ch, H, W), 5, "cq")
        h = np.tanh(_synth.normal((B, Ch, H, W), 5, "h"))
        azr = G(np.concatenate([az, ar], 1))
        ctx = torch.zeros(B, 3 * Ch, H, W, device=DEV)
        ctx[:, :Ch], ctx[:, Ch:2 * Ch], ctx[:, 2 * Ch:] = G(cz), G(cr), G(cq)
        gcz, gcr, gcq = ctx.split(Ch, dim=1)
        gh = G(h)
        z = torch.empty(B, Ch, H, W, device=DEV)
        hx = torch.zeros(B, Ch + 5, H, W, device=DEV)
        st, dev = _ffi.stream_of(gh), 0
        _ffi.check(L.dkt_gru_gate_zr(azr.data_ptr(), gcz.data_ptr(), gcz.stride(0), gcr.data_ptr(), gcr.stride(0),
                                     gh.data_ptr(), gh.stride(0), z.data_ptr(), hx.data_ptr(), hx.stride(0),
                                     B, Ch, HW, dev, st), "zr")
        wz, wrh = c_oracle.gru_gate_zr(az, ar, cz, cr, h)
        assert maxabs(z, wz) <= 3e-7 and maxabs(hx[:, :Ch], wrh) <= 3e-7
        assert float(hx[:, Ch:].abs().max()) == 0.0
        out = torch.empty(B, Ch, H, W, device=DEV)
        _ffi.check(L.dkt_gru_gate_out(G(aq).data_ptr(), gcq.data_ptr(), gcq.stride(0), z.data_ptr(),
                                      gh.data_ptr(), gh.stride(0), out.data_ptr(), out.stride(0),
                                      B, Ch, HW, dev, st), "out")
        assert maxabs(out, c_oracle.gru_gate_out(aq, cq, z.cpu().numpy(), h)) <= 3e-7


def _make_block(c):
    from dkt_stereo_amd.update import BasicMultiUpdateBlock, BasicMultiUpdateBlockIGEV
    cfg = _cases.update_cfg(c)
    cls = BasicMultiUpdateBlockIGEV if c["igev"] else BasicMultiUpdateBlock
    blk = cls(SimpleNamespace(**cfg), hidden_dims=cfg["hidden_dims"])
    shapes = {"update_block." + k: tuple(v.shape) for k, v in blk.state_dict().items()}
    sd = _synth.torch_state_dict(shapes, c["seed"])
    blk.load_state_dict({k[len("update_block."):]: v for k, v in sd.items()}, strict=True)
    return blk.to(DEV).eval(), sd


@pytest.mark.parametrize("name", list(_cases.UPDATE_CASES))
@torch.no_grad()
def test_update_block(name, golden):
    c = _cases.UPDATE_CASES[name]
    blk, _ = _make_block(c)
    net, inp, corr, flow = _cases.update_inputs(c)
    n = c["n"]
    gnet = [G(x) for x in net]
    caller_list = gnet
    ginp = [list(G(x).split(128, dim=1)) for x in inp]
    if c["igev"]:
        rnet, rmask, rdelta = blk(gnet, ginp, G(corr), G(flow), iter16=(n == 3), iter08=(n >= 2))
    else:
        rnet, rmask, rdelta = blk(gnet, ginp, G(corr), G(flow), iter32=(n == 3), iter16=(n >= 2))
    assert rnet is caller_list            # the reference mutates and returns the caller's list
    g = golden("update")
    for i in range(3):
        assert maxabs(rnet[i], g["%s/net%d" % (name, i)]) <= 1e-5
    assert maxabs(rmask[:, :, ::2, ::2], g[name + "/mask"]) <= 5e-5
    assert maxabs(rdelta, g[name + "/delta"]) <= 5e-5
    enc = blk.encoder(G(flow), G(corr))
    assert maxabs(enc[:, ::4], g[name + "/motion"]) <= 1e-5
    # update=False returns only the list; need_mask=False skips the mask head
    only = blk([G(x) for x in net], ginp, iter08=False, iter16=(n >= 2), update=False) if not c["igev"] else \
        blk([G(x) for x in net], ginp, iter04=False, iter08=(n >= 2), update=False)
    assert isinstance(only, list) and len(only) == 3
    kw = dict(iter16=(n == 3), iter08=(n >= 2)) if c["igev"] else dict(iter32=(n == 3), iter16=(n >= 2))
    _, nomask, d2 = blk([G(x) for x in net], ginp, G(corr), G(flow), need_mask=False, **kw)
    assert nomask is None and maxabs(d2, rdelta) <= 1e-5   # (vendor conv may pick another algorithm per call)


# ---------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------
def _raft(device=DEV, **over):
    from dkt_stereo_amd.raft_stereo import RAFTStereo, make_args
    m = RAFTStereo(make_args(**over))
    sd = _synth.torch_state_dict(_synth.shapes_of(m), _cases.E2E_WEIGHT_SEED)
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval(), sd


@pytest.mark.parametrize("name", list(_cases.E2E_CASES))
@torch.no_grad()
def test_raft_stereo_end_to_end(name, golden):
    """Final disparity within 1e-3 max-abs of the reference (north_star) -- cases
    include BASELINE.json configs[0] (256x512, 8 iters) and a 32-iteration run."""
    c = _cases.E2E_CASES[name]
    model, _ = _raft()
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    lo, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
    g = golden("raft_e2e")
    s = int(g[name + "/stride"])
    d_up = maxabs(up[:, :, ::s, ::s], g[name + "/flow_up"])
    d_lo = maxabs(lo[:, :1], g[name + "/flow_lo"])
    epe = float(np.mean(np.abs(up[:, :, ::s, ::s].cpu().numpy() - g[name + "/flow_up"])))
    print("%s: max|d_up| %.3e max|d_lo| %.3e EPE %.3e" % (name, d_up, d_lo, epe))
    assert d_up <= 1e-3 and d_lo <= 1e-3 and epe <= 1e-3


@torch.no_grad()
def test_raft_stereo_alt_and_batch(golden):
    c = _cases.E2E_CASES["64x128_it12"]
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    model, _ = _raft(corr_implementation="alt")
    _, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
    assert maxabs(up, golden("raft_e2e")["64x128_it12/flow_up_alt"]) <= 1e-3
    # batch of two different pairs == the two pairs run alone (pairs are independent)
    model, _ = _raft()
    j1, j2 = _synth.image_pair(5, 1, c["H"], c["W"], 20)
    both = model(G(np.concatenate([i1, j1])), G(np.concatenate([i2, j2])), iters=6, test_mode=True)[1]
    a = model(G(i1), G(i2), iters=6, test_mode=True)[1]
    b = model(G(j1), G(j2), iters=6, test_mode=True)[1]
    assert maxabs(both[:1], a) <= 1e-3 and maxabs(both[1:], b) <= 1e-3


@torch.no_grad()
def test_igev_loop(golden):
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    c = _cases.IGEV_LOOP_CASES["small"]
    s = c["seed"]
    blk, _ = _make_block(dict(seed=s, igev=True, n=3))
    m1, m2, geo, disp, coords = _cases.geo_inputs(dict(c, L=2, r=4))
    B, H, W = c["B"], c["H"], c["W"]
    net = [G(np.tanh(_synth.normal((B, 128, H >> i, W >> i), s, "net%d" % i))) for i in range(3)]
    inp = [list(G(_synth.normal((B, 384, H >> i, W >> i), s, "inp%d" % i, scale=0.5)).split(128, dim=1)) for i in range(3)]
    geo_fn = Combined_Geo_Encoding_Volume(G(m1), G(m2), G(geo), radius=4, num_levels=2)
    d = G(np.abs(disp))
    for _ in range(c["iters"]):        # igev_stereo.py:199-210
        feat = geo_fn(d, G(coords))
        net, mask, delta = blk(net, inp, feat, d, iter16=True, iter08=True)
        d = d + delta
    g = golden("igev_loop")
    assert maxabs(d, g["small/disp"]) <= 1e-3
    assert maxabs(mask, g["small/mask"]) <= 1e-3


@pytest.mark.parametrize("c8", [False, True])
@torch.no_grad()
def test_igev_iterate_graph_pipeline_equals_plain_loop(golden, c8, monkeypatch):
    """dkt_stereo_amd.igev_loop.igev_iterate matches the reference fixture and the reference-order loop: the round-2 loop
    (HIP graph + GRUs pipelined across iterations on two streams; c8 = False) bit for bit, the default loop (loop_c8, every
    image size since the end of round 4; c8 = True) within the split-fp16 class -- the same arithmetic in another
    accumulation order -- and bit for bit between its own eager first call and the replays."""
    from dkt_stereo_amd import igev_loop
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    from dkt_stereo_amd.igev_loop import _plain, igev_iterate
    monkeypatch.setattr(igev_loop, "USE_C8", c8)
    c = _cases.IGEV_LOOP_CASES["small"]
    s = c["seed"]
    blk, _ = _make_block(dict(seed=s, igev=True, n=3))
    m1, m2, geo, disp, coords = _cases.geo_inputs(dict(c, L=2, r=4))
    B, H, W = c["B"], c["H"], c["W"]
    net = [G(np.tanh(_synth.normal((B, 128, H >> i, W >> i), s, "net%d" % i))) for i in range(3)]
    inp = [list(G(_synth.normal((B, 384, H >> i, W >> i), s, "inp%d" % i, scale=0.5)).split(128, dim=1)) for i in range(3)]
    geo_fn = Combined_Geo_Encoding_Volume(G(m1), G(m2), G(geo), radius=4, num_levels=2)
    d0 = G(np.abs(disp))
    g = golden("igev_loop")

    def same(a, b):
        return torch.equal(a, b) if not c8 else maxabs(a, b) <= 2e-4

    want_d, want_m, want_net = _plain(blk, geo_fn, d0, G(coords), [t.clone() for t in net], inp, c["iters"])
    assert maxabs(want_d, g["small/disp"]) <= 1e-3 and maxabs(want_m, g["small/mask"]) <= 1e-3
    cache = {}
    first = None
    for _ in range(2):                               # second call replays the cached graph
        got_d, got_m, got_net = igev_iterate(blk, geo_fn, d0, G(coords), [t.clone() for t in net], inp, c["iters"], cache=cache)
        assert same(got_d, want_d) and same(got_m, want_m)
        assert same(got_net[0], want_net[0]) and same(got_net[1], want_net[1])
        assert maxabs(got_d, g["small/disp"]) <= 1e-3 and maxabs(got_m, g["small/mask"]) <= 1e-3
        if first is not None:
            assert torch.equal(got_d, first[0]) and torch.equal(got_m, first[1])
        first = (got_d, got_m)
    # more iterations than the fixture has: still the plain loop's result
    w7 = _plain(blk, geo_fn, d0, G(coords), [t.clone() for t in net], inp, 7)
    g7 = igev_iterate(blk, geo_fn, d0, G(coords), [t.clone() for t in net], inp, 7, cache=cache)
    assert same(g7[0], w7[0]) and same(g7[1], w7[1])


# ---------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (no oracle at this size)
# ---------------------------------------------------------------------------------
@torch.no_grad()
def test_full_size_corr_properties():
    """cfg2: 736x1248 -> 184x312, C=256, L=4, r=4."""
    from dkt_stereo_amd.corr import CorrBlock1D
    B, C, H, W = 1, 256, 184, 312
    f1, f2 = (G(_synth.normal((B, C, H, W), 3, n)) for n in ("a", "b"))
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    p0 = blk.corr_pyramid[0].view(B * H * W, W)
    # (1) spot rows against an fp64 contraction
    for (h, w1) in ((0, 0), (91, 155), (183, 311)):
        want = (f1[0, :, h, w1].double()[:, None] * f2[0, :, h, :].double()).sum(0) / 16.0
        assert maxabs(p0[h * W + w1].double(), want) <= 2e-5
    # (2) linearity in fmap1: corr(a1+a2, b) == corr(a1,b)+corr(a2,b) up to round-off
    f1b = G(_synth.normal((B, C, H, W), 4, "a2"))
    lhs = CorrBlock1D.corr(f1 + f1b, f2).view(-1, W)
    rhs = p0 + CorrBlock1D.corr(f1b, f2).view(-1, W)
    assert float((lhs - rhs).abs().max()) <= 5e-5
    # (3) every pooled level is exactly the pairwise mean of the level above
    for i in range(1, 4):
        up = blk.corr_pyramid[i - 1].view(B * H * W, -1)
        wi = up.shape[1] // 2
        assert torch.equal(blk.corr_pyramid[i].view(B * H * W, -1), (up[:, 0:2 * wi:2] + up[:, 1:2 * wi:2]) * 0.5)
    # (4) lookup at integer coordinates returns volume entries / zero padding exactly
    coords = torch.zeros(B, 2, H, W, device=DEV)
    xs = torch.arange(W, device=DEV).float() - 7.0
    coords[:, 0] = xs.view(1, 1, W)
    out = blk(coords)
    w1 = torch.arange(W, device=DEV)
    for k in range(9):
        idx = w1 - 7 + (k - 4)
        ok = (idx >= 0) & (idx < W)
        want = torch.where(ok.view(1, W), p0.view(H, W, W)[:, w1, idx.clamp(0, W - 1)], torch.zeros((), device=DEV))
        # the reference's coordinate round trip leaves ~W*2^-23 of weight on the neighbour tap
        assert float((out[0, k] - want).abs().max()) <= 5e-4
    # (5) a checksum of checksums is reproducible run to run (deterministic kernel)
    again = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    for a, b in zip(blk.corr_pyramid, again.corr_pyramid):
        assert torch.equal(a, b)
    assert torch.equal(again(coords), out)


@torch.no_grad()
def test_full_size_volume_properties():
    """cfg3 IGEV gwc (96ch, G=8, D=48 @184x312) and cfg5 GwcNet (320ch, G=40 @136x240)."""
    from dkt_stereo_amd.submodule import build_concat_volume, build_gwc_volume
    for (C, G_, H, W) in ((96, 8, 184, 312), (320, 40, 136, 240)):
        a, b = (G(_synth.normal((1, C, H, W), 8, n)) for n in ("a", "b"))
        vol = build_gwc_volume(a, b, 48, G_)
        assert vol.shape == (1, G_, 48, H, W)
        cpg = C // G_
        for d in (0, 1, 17, 47):
            want = (a[..., d:] * b[..., :W - d]).view(1, G_, cpg, H, W - d).mean(2)
            assert float((vol[:, :, d, :, d:] - want).abs().max()) <= 2e-6
            if d:
                assert float(vol[:, :, d, :, :d].abs().max()) == 0.0
    a, b = (G(_synth.normal((1, 12, 136, 240), 9, n)) for n in ("a", "b"))
    cv = build_concat_volume(a, b, 48)
    for d in (0, 5, 47):
        assert torch.equal(cv[:, :12, d, :, d:], a[..., d:])
        assert torch.equal(cv[:, 12:, d, :, d:], b[..., :240 - d])
        if d:
            assert float(cv[:, :, d, :, :d].abs().max()) == 0.0


@torch.no_grad()
def test_full_size_geo_lookup_properties():
    """cfg3: geo (1,8,48,184,312); integer disparities pick volume planes exactly."""
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    B, C, D, H, W = 1, 8, 48, 184, 312
    m1, m2 = (G(_synth.normal((B, 96, H, W), 6, n)) for n in ("a", "b"))
    geo = G(_synth.normal((B, C, D, H, W), 6, "geo"))
    fn = Combined_Geo_Encoding_Volume(m1, m2, geo, num_levels=2, radius=4)
    disp = torch.full((B, 1, H, W), 10.0, device=DEV)
    coords = torch.arange(W, device=DEV).float().view(1, 1, W, 1).repeat(B, H, 1, 1)
    out = fn(disp, coords)
    assert out.shape == (B, 162, H, W)
    for c in (0, 7):
        for k in range(9):
            assert float((out[:, c * 9 + k] - geo[:, c, 10 + k - 4]).abs().max()) <= 1e-5
    half = (geo[:, :, 0::2] + geo[:, :, 1::2]) * 0.5
    assert torch.equal(fn.geo_volume_pyramid[1], half)
    for k in range(9):   # level 1 (offset 81): disp/2 = 5
        assert float((out[:, 81 + k] - half[:, 0, 5 + k - 4]).abs().max()) <= 1e-5


@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
@torch.no_grad()
def test_skewed_pyramid_lookup_is_bit_identical(name):
    """The diagonal-major pyramid (dkt_corr1d_skew / dkt_corr1d_lookup_skew) is a pure
    re-indexing: same values, same taps -> the same bits as the reference-layout lookup,
    for smooth, random, integral and far out-of-range coordinates and for W1 != W2."""
    from dkt_stereo_amd.corr import CorrBlock1D, _lookup
    c = _cases.CORR_CASES[name]
    f1, f2, coords = _cases.corr_inputs(c)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    assert blk._skew is not None and CorrBlock1D.lookup_layout == "skew"
    rows = _lookup(blk.corr_pyramid, G(coords), c["r"], c["W2"])
    for cc in (coords, _synth.coords(77, c["B"], c["H"], c["W"], spread=3.0)):
        assert torch.equal(blk(G(cc)), _lookup(blk.corr_pyramid, G(cc), c["r"], c["W2"]))
    assert torch.equal(blk(G(coords)), rows)
    # the skew itself: S[row][s][w1] == P[row*W1 + w1][(s + (w1 >> i)) % W2_i]
    B, H, W1 = c["B"], c["H"], c["W"]
    for i, (p, s) in enumerate(zip(blk.corr_pyramid, blk._skew)):
        wi = c["W2"] >> i
        P = p.view(B * H, W1, wi).cpu()
        S = s.view(B * H, wi, -1)[:, :, :W1].cpu()
        w1 = torch.arange(W1)
        for sv in (0, 1, wi - 1, wi // 2):
            col = (sv + (w1 >> i)) % wi
            assert torch.equal(S[:, sv, :], P[:, w1, col])


@pytest.mark.parametrize("W", [312, 77, 640, 1000])
@torch.no_grad()
def test_skew_lookup_fast_division_is_exact(W):
    """dkt_corr1d_lookup_skew evaluates the sampler's 2x/(W-1) with a reciprocal + two fma
    corrections; dkt_corr1d_lookup uses the IEEE division.  Bit-identical outputs over
    ~10^8 divisions per width (random, integral, half-integral and out-of-range coordinates,
    4 pyramid levels = 4 divisors per width)."""
    from dkt_stereo_amd.corr import CorrBlock1D, _lookup
    H, C = 48, 8
    g = torch.Generator(device=DEV).manual_seed(1000 + W)
    f1 = torch.randn(1, C, H, W, device=DEV, generator=g)
    f2 = torch.randn(1, C, H, W, device=DEV, generator=g)
    blk = CorrBlock1D(f1, f2, num_levels=4, radius=4)
    assert blk._skew is not None
    base = torch.arange(W, device=DEV, dtype=torch.float32).view(1, 1, 1, W).expand(1, 1, H, W)
    for rep in range(24):
        r = torch.rand(1, 1, H, W, device=DEV, generator=g)
        if rep % 4 == 0:
            x = base - (r * 64).floor()                         # integral
        elif rep % 4 == 1:
            x = base - (r * 128).floor() * 0.5                  # half-integral
        elif rep % 4 == 2:
            x = (r * (W + 40) - 20)                             # anywhere, incl. out of range
        else:
            x = base - r * 60 * (1 + rep)                       # smooth-ish, far left for large rep
        coords = torch.cat([x, torch.zeros_like(x)], 1).contiguous()
        assert torch.equal(blk(coords), _lookup(blk.corr_pyramid, coords, 4, W))



@pytest.mark.parametrize("name", list(_cases.PCV_CASES))
@torch.no_grad()
def test_pcvnet_corr_block(name, golden, c_oracle):
    """meta_arch/pcvnet/corr.py through dkt_pool_rows / dkt_pcv_lookup: pyramid levels > 0 and
    the lookup are bit-exact on identical level-0 data; level 0 within contraction round-off."""
    from dkt_stereo_amd.pcvnet_corr import CorrBlock1D
    c = _cases.PCV_CASES[name]
    f1, f2, coords, sigma = _cases.pcv_inputs(c)
    g = golden("pcv_cgi")
    blk = CorrBlock1D(G(f1), G(f2), sample_num=c["S"], num_levels=c["L"], downsample=c["downsample"])
    assert len(blk.corr_pyramid) == c["L"] and blk.compress_factor == (4 if c["downsample"] == 2 else 2)
    want0 = g["pcv/%s/pyr0" % name]
    scale = max(float(np.abs(want0).max()), 1.0)
    got = [p.view(p.shape[0], -1).cpu().numpy() for p in blk.corr_pyramid]
    assert maxabs(got[0], want0) <= 4e-6 * scale
    own = c_oracle.pcv_pyramid(got[0], c["L"], blk.compress_factor)          # pooling of OUR level 0
    for i in range(1, c["L"]):
        assert np.array_equal(got[i], own[i])
        assert maxabs(got[i], g["pcv/%s/pyr%d" % (name, i)]) <= 4e-6 * scale
    out = blk(G(coords), G(sigma)).cpu().numpy()
    assert out.shape == g["pcv/%s/lookup" % name].shape
    assert np.array_equal(out, c_oracle.pcv_lookup(got, coords, sigma, c["S"], blk.compress_factor))
    assert maxabs(out, g["pcv/%s/lookup" % name]) <= 8e-6 * scale
    # lookup alone on the reference's own pyramid: bit-exact
    blk.corr_pyramid = [G(g["pcv/%s/pyr%d" % (name, i)]).view(-1, 1, 1, got[i].shape[1]) for i in range(c["L"])]
    assert np.array_equal(blk(G(coords), G(sigma)).cpu().numpy(), g["pcv/%s/lookup" % name])


@pytest.mark.parametrize("name", list(_cases.NORMCORR_CASES))
@torch.no_grad()
def test_cgi_norm_correlation_volumes(name, golden, c_oracle):
    """meta_arch/cgi/submodule.py:143-180 through dkt_group_l2norm + dkt_gwc_volume."""
    from dkt_stereo_amd.submodule import build_gwc_volume_norm, build_norm_correlation_volume
    c = _cases.NORMCORR_CASES[name]
    a, b = _cases.volume_inputs(c)
    g = golden("pcv_cgi")
    from dkt_stereo_amd import submodule as sm
    assert maxabs(build_gwc_volume_norm(G(a), G(b), c["D"], c["G"]), g["normcorr/%s/gwc_norm" % name]) <= 1e-6
    with sm.gwc_mode("exact"):
        vol = build_gwc_volume_norm(G(a), G(b), c["D"], c["G"]).cpu().numpy()
    assert maxabs(vol, g["normcorr/%s/gwc_norm" % name]) <= 1e-6
    assert np.array_equal(vol, c_oracle.gwc_volume_norm(a, b, c["D"], c["G"]))    # same order -> bit exact
    if c["G"] == 1:
        v1 = build_norm_correlation_volume(G(a), G(b), c["D"]).cpu().numpy()
        assert v1.shape == (c["B"], 1, c["D"], c["H"], c["W"])
        assert maxabs(v1, g["normcorr/%s/norm_corr" % name]) <= 1e-6



@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
def test_corr_backward(name, golden, c_oracle):
    """Autograd through CorrBlock1D (dkt_corr1d_lookup_bwd / dkt_corr1d_pool_bwd + GEMMs) against
    the reference's autograd (fixture) and the C oracle: the scatter is bit-exact on identical
    inputs; feature gradients within contraction round-off."""
    from dkt_stereo_amd.corr import CorrBlock1D
    c = _cases.CORR_CASES[name]
    f1, f2, coords = _cases.corr_inputs(c)
    g = golden("corr_bwd")
    K = 2 * c["r"] + 1
    R = _synth.normal((c["B"], c["L"] * K, c["H"], c["W"]), c["seed"], "gout")
    a, b = G(f1).requires_grad_(True), G(f2).requires_grad_(True)
    blk = CorrBlock1D(a, b, num_levels=c["L"], radius=c["r"])
    assert all(p.requires_grad for p in blk.corr_pyramid)
    out = blk(G(coords))
    with torch.no_grad():
        ref_fwd = CorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])(G(coords))
    assert torch.equal(out.detach(), ref_fwd)                       # same forward kernels
    glv = torch.autograd.grad(out, blk.corr_pyramid, G(R), retain_graph=True)
    widths = [c["W2"] >> i for i in range(c["L"])]
    cg = c_oracle.corr1d_lookup_bwd(R, coords, c["r"], widths, c["B"] * c["H"] * c["W"])
    for i in range(c["L"]):
        got = glv[i].view(glv[i].shape[0], -1).cpu().numpy()
        assert np.array_equal(got, cg[i])                            # per-level scatter: bit exact
    gf1, gf2 = torch.autograd.grad(out, [a, b], G(R))
    for got, key in ((gf1, "gf1"), (gf2, "gf2")):
        want = g["%s/%s" % (name, key)]
        assert maxabs(got, want) <= 4e-6 * max(float(np.abs(want).max()), 1.0)
    # coordinates must be detached, like the reference's callers do
    with pytest.raises(Exception):
        blk(G(coords).requires_grad_(True))


def test_corr_backward_accumulates_over_lookups():
    """Several lookups on one pyramid (the GRU loop): gradients of the feature maps add up."""
    from dkt_stereo_amd.corr import CorrBlock1D
    c = _cases.CORR_CASES["small"]
    f1, f2, coords = _cases.corr_inputs(c)
    a, b = G(f1).requires_grad_(True), G(f2).requires_grad_(True)
    blk = CorrBlock1D(a, b, num_levels=c["L"], radius=c["r"])
    c1, c2 = G(coords), G(coords) - 1.75
    (blk(c1).sum() + 2.0 * blk(c2).sum()).backward()
    both = a.grad.clone()
    a.grad = None; b.grad = None
    blk = CorrBlock1D(a, b, num_levels=c["L"], radius=c["r"])
    blk(c1).sum().backward()
    g1 = a.grad.clone(); a.grad = None
    blk = CorrBlock1D(a, b, num_levels=c["L"], radius=c["r"])
    (2.0 * blk(c2).sum()).backward()
    assert float((both - (g1 + a.grad)).abs().max()) <= 2e-5 * float(both.abs().max())



@pytest.mark.parametrize("name", list(_cases.UPSAMPLE_CASES))
@torch.no_grad()
def test_convex_upsample(name, golden, c_oracle):
    """dkt_convex_upsample vs RAFTStereo.upsample_flow (fixture) and the C oracle."""
    from dkt_stereo_amd.raft_stereo import RAFTStereo, make_args
    c = _cases.UPSAMPLE_CASES[name]
    flow, mask, f = _cases.upsample_inputs(c)
    fake = type("M", (), {"args": make_args(n_downsample=c["nd"])})()
    got = RAFTStereo.upsample_flow(fake, G(flow), G(mask)).cpu().numpy()
    want = golden("upsample")["convex/" + name]
    assert got.shape == want.shape
    assert maxabs(got, want) <= 4e-6 * float(np.abs(want).max())
    assert maxabs(got, c_oracle.convex_upsample(flow, mask, f)) <= 4e-6 * float(np.abs(want).max())   # expf ulps


@pytest.mark.parametrize("name", list(_cases.CONTEXT_UP_CASES))
@torch.no_grad()
def test_context_upsample(name, golden, c_oracle):
    from dkt_stereo_amd.submodule import context_upsample
    c = _cases.CONTEXT_UP_CASES[name]
    disp, wts = _cases.context_up_inputs(c)
    got = context_upsample(G(disp), G(wts)).cpu().numpy()
    want = golden("upsample")["context/" + name]
    assert got.shape == want.shape and maxabs(got, want) <= 2e-6 * float(np.abs(want).max())
    assert np.array_equal(got, c_oracle.context_upsample(disp, wts))       # same order -> bit exact



@pytest.mark.parametrize("name", list(_cases.GEO_CASES))
def test_geo_volume_backward(name, golden, c_oracle):
    """Autograd through Combined_Geo_Encoding_Volume (dkt_geo_lookup_bwd / dkt_geo_pool_bwd / corr backward)
    against the reference's autograd: geometry-volume gradient bit-exact, feature gradients within round-off."""
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    c = _cases.GEO_CASES[name]
    m1, m2, geo, disp, coords = _cases.geo_inputs(c)
    K = 2 * c["r"] + 1
    R = _synth.normal((c["B"], c["L"] * K * (c["C"] + 1), c["H"], c["W"]), c["seed"], "ggeo")
    a, b, gv = G(m1).requires_grad_(True), G(m2).requires_grad_(True), G(geo).requires_grad_(True)
    vol = Combined_Geo_Encoding_Volume(a, b, gv, num_levels=c["L"], radius=c["r"])
    out = vol(G(disp), G(coords))
    with torch.no_grad():
        ref_fwd = Combined_Geo_Encoding_Volume(G(m1), G(m2), G(geo), num_levels=c["L"], radius=c["r"])(G(disp), G(coords))
    assert torch.equal(out.detach(), ref_fwd)
    ga, gb, gg = torch.autograd.grad(out, [a, b, gv], G(R))
    g = golden("geo_bwd")
    assert np.array_equal(gg.cpu().numpy(), g[name + "/ggeo"])                      # scatter + pooled chain: bit exact
    for got, key in ((ga, "gm1"), (gb, "gm2")):
        want = g["%s/%s" % (name, key)]
        assert maxabs(got, want) <= 4e-6 * max(float(np.abs(want).max()), 1.0)
    with pytest.raises(Exception):
        vol(G(disp).requires_grad_(True), G(coords))
