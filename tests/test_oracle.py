"""CPU suite: both oracles (oracle/torch_oracle.py, oracle/dkt_oracle.c) against the
committed outputs of the reference (tests/golden/*.npz, made by make_golden.py).
Tolerances: 0 where the restatement executes the same primitives in the same
order (bit-exact on the machine that generated the fixtures; a different host
CPU may change BLAS/vector-width summation order, hence the tiny non-zero
bounds); fp32 round-off class where the reference's summation order is
unspecified."""
import json
import os

import numpy as np
import pytest
import torch

import _cases
import _synth
from oracle import torch_oracle as to

T = torch.from_numpy


def maxabs(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def flat(p):
    return p.reshape(p.shape[0], -1)


def test_sampler_bit_exact(golden, c_oracle):
    g = golden("sampler")
    rng = _synth.rng(99, "sampler")
    rows = rng.standard_normal((20000, 39)).astype(np.float32)
    got = np.array([c_oracle.bilinear_1d(rows[i], float(g["x"][i])) for i in range(2000)], np.float32)
    assert np.array_equal(got, g["out"])


@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
def test_corr_oracles(name, golden, c_oracle):
    c = _cases.CORR_CASES[name]
    g = golden("corr")
    f1, f2, coords = _cases.corr_inputs(c)
    s = int(g[name + "/pyr_stride"])
    scale = max(1.0, float(np.abs(g[name + "/pyr0"]).max()))
    with torch.no_grad():
        tp = to.corr1d_pyramid(T(f1), T(f2), c["L"])
        tl = to.corr1d_lookup(tp, T(coords), c["r"]).numpy()
        ta = to.corr1d_lookup_alt(T(f1), T(f2), T(coords), c["L"], c["r"]).numpy()
    cp = c_oracle.corr1d_build(f1, f2, c["L"])
    for i in range(c["L"]):
        assert maxabs(flat(tp[i].numpy())[::s], g["%s/pyr%d" % (name, i)]) <= 4e-6 * scale
        assert maxabs(cp[i][::s], g["%s/pyr%d" % (name, i)]) <= 4e-6 * scale
    assert maxabs(tl, g[name + "/lookup"]) <= 1e-5 * scale
    assert maxabs(ta, g[name + "/alt"]) <= 1e-5 * scale
    # C oracle end to end (own pyramid -> lookup) and the alt variant
    assert maxabs(c_oracle.corr1d_lookup(cp, coords, c["r"]), g[name + "/lookup"]) <= 1e-5 * scale
    assert maxabs(c_oracle.corr1d_lookup_alt(f1, f2, coords, c["L"], c["r"]), g[name + "/alt"]) <= 1e-5 * scale
    if s == 1:  # sampler arithmetic alone, on the reference's own pyramid: bit exact
        pyr = [g["%s/pyr%d" % (name, i)] for i in range(c["L"])]
        assert np.array_equal(c_oracle.corr1d_lookup(pyr, coords, c["r"]), g[name + "/lookup"])
        pooled = c_oracle.pool_pyramid(pyr[0], c["L"])
        for a, b in zip(pooled, pyr):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
def test_cosine_oracle(name, golden):
    c = _cases.CORR_CASES[name]
    g = golden("corr")
    f1, f2, coords = _cases.corr_inputs(c)
    s = int(g[name + "/pyr_stride"])
    with torch.no_grad():
        vol = to.corr1d_volume_cosine(T(f1), T(f2))
    n = vol.shape[0] * vol.shape[1] * vol.shape[2]
    assert maxabs(vol.reshape(n, -1).numpy()[::s], g[name + "/cos0"]) <= 2e-6


@pytest.mark.parametrize("name", list(_cases.GEO_CASES))
def test_geo_oracles(name, golden, c_oracle):
    c = _cases.GEO_CASES[name]
    g = golden("geo")
    m1, m2, geo, disp, coords = _cases.geo_inputs(c)
    want = g[name + "/lookup"]
    scale = max(1.0, float(np.abs(want).max()))
    with torch.no_grad():
        gp, ip = to.geo_pyramids(T(m1), T(m2), T(geo), c["L"])
        tl = to.geo_lookup(gp, ip, T(disp), T(coords), c["r"]).numpy()
    assert maxabs(tl, want) <= 1e-5 * scale
    cgp, cip = c_oracle.geo_pyramids(m1, m2, geo, c["L"])
    assert maxabs(cip[0], g[name + "/init0"]) <= 4e-6 * scale
    assert maxabs(c_oracle.geo_lookup(cgp, cip, disp, coords, c["C"], c["r"]), want) <= 1e-5 * scale


@pytest.mark.parametrize("name", list(_cases.GWC_CASES))
def test_gwc_oracles(name, golden, c_oracle):
    c = _cases.GWC_CASES[name]
    a, b = _cases.volume_inputs(c)
    want = golden("volumes")["gwc/" + name]
    with torch.no_grad():
        assert maxabs(to.gwc_volume(T(a), T(b), c["D"], c["G"]).numpy(), want) <= 2e-6
    assert maxabs(c_oracle.gwc_volume(a, b, c["D"], c["G"]), want) <= 2e-6


@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
def test_corr_backward_oracle(name, golden, c_oracle):
    """C restatement of the lookup / pyramid backward against the reference's autograd
    (tests/golden/corr_bwd.npz): level gradients bit-exact, feature gradients within the
    contraction's round-off."""
    c = _cases.CORR_CASES[name]
    f1, f2, coords = _cases.corr_inputs(c)
    g = golden("corr_bwd")
    K = 2 * c["r"] + 1
    R = _synth.normal((c["B"], c["L"] * K, c["H"], c["W"]), c["seed"], "gout")
    widths = [c["W2"] >> i for i in range(c["L"])]
    cg = c_oracle.corr1d_lookup_bwd(R, coords, c["r"], widths, c["B"] * c["H"] * c["W"])
    for i in range(c["L"]):
        assert np.array_equal(c_oracle.corr1d_pool_bwd(cg[i:], 1.0), g["%s/glevel%d" % (name, i)])
    g0 = c_oracle.corr1d_pool_bwd(cg, float(np.sqrt(np.float32(c["C"]))))
    gf1, gf2 = c_oracle.corr1d_build_bwd(g0, f1, f2)
    assert maxabs(gf1, g["%s/gf1" % name]) <= 4e-6 * max(float(np.abs(g["%s/gf1" % name]).max()), 1.0)
    assert maxabs(gf2, g["%s/gf2" % name]) <= 4e-6 * max(float(np.abs(g["%s/gf2" % name]).max()), 1.0)


@pytest.mark.parametrize("name", list(_cases.UPSAMPLE_CASES))
def test_convex_upsample_oracles(name, golden, c_oracle):
    c = _cases.UPSAMPLE_CASES[name]
    flow, mask, f = _cases.upsample_inputs(c)
    want = golden("upsample")["convex/" + name]
    with torch.no_grad():
        assert maxabs(to.convex_upsample(T(flow), T(mask), f).numpy(), want) <= 2e-6 * float(np.abs(want).max())
    assert maxabs(c_oracle.convex_upsample(flow, mask, f), want) <= 2e-6 * float(np.abs(want).max())


@pytest.mark.parametrize("name", list(_cases.CONTEXT_UP_CASES))
def test_context_upsample_oracles(name, golden, c_oracle):
    c = _cases.CONTEXT_UP_CASES[name]
    disp, wts = _cases.context_up_inputs(c)
    want = golden("upsample")["context/" + name]
    with torch.no_grad():
        assert maxabs(to.context_upsample(T(disp), T(wts)).numpy(), want) <= 2e-6 * float(np.abs(want).max())
    assert maxabs(c_oracle.context_upsample(disp, wts), want) <= 2e-6 * float(np.abs(want).max())


@pytest.mark.parametrize("name", list(_cases.GEO_CASES))
def test_geo_backward_oracle(name, golden, c_oracle):
    """C restatement of the geometry-volume lookup backward against the reference's autograd
    (tests/golden/geo_bwd.npz): volume gradient bit-exact, feature gradients within round-off."""
    c = _cases.GEO_CASES[name]
    m1, m2, geo, disp, coords = _cases.geo_inputs(c)
    g = golden("geo_bwd")
    K = 2 * c["r"] + 1
    R = _synth.normal((c["B"], c["L"] * K * (c["C"] + 1), c["H"], c["W"]), c["seed"], "ggeo")
    B, C, D, H, W = geo.shape
    cg, ci = c_oracle.geo_lookup_bwd(R, disp, coords, C, D, c["W"], c["L"], c["r"])
    tg = c_oracle.corr1d_pool_bwd(cg, 1.0).reshape(B, H, W, C, D).transpose(0, 3, 4, 1, 2)
    assert np.array_equal(tg, g[name + "/ggeo"])
    f1, f2 = c_oracle.corr1d_build_bwd(c_oracle.corr1d_pool_bwd(ci, 1.0), m1, m2)
    for got, key in ((f1, "gm1"), (f2, "gm2")):
        want = g["%s/%s" % (name, key)]
        assert maxabs(got, want) <= 4e-6 * max(float(np.abs(want).max()), 1.0)


@pytest.mark.parametrize("name", list(_cases.PCV_CASES))
def test_pcv_oracles(name, golden, c_oracle):
    """PCVNet correlation block (meta_arch/pcvnet/corr.py): pooling by the compress factor and
    the sigma-spaced lookup -- both oracles reproduce the reference bit for bit."""
    c = _cases.PCV_CASES[name]
    f1, f2, coords, sigma = _cases.pcv_inputs(c)
    g = golden("pcv_cgi")
    pyr = [g["pcv/%s/pyr%d" % (name, i)] for i in range(c["L"])]
    with torch.no_grad():
        tp, factor = to.pcv_pyramid(T(f1), T(f2), c["L"], c["downsample"])
        scale = max(float(np.abs(pyr[0]).max()), 1.0)
        for i in range(c["L"]):
            assert maxabs(flat(tp[i].numpy()), pyr[i]) <= 4e-6 * scale          # BLAS order of the host
        ref_pyr = [T(p).view(p.shape[0], 1, 1, -1) for p in pyr]
        assert maxabs(to.pcv_lookup(ref_pyr, T(coords), T(sigma), c["S"], factor).numpy(), g["pcv/%s/lookup" % name]) == 0.0
    cp = c_oracle.pcv_pyramid(pyr[0], c["L"], factor)
    for i in range(c["L"]):
        assert np.array_equal(cp[i], pyr[i])
    assert np.array_equal(c_oracle.pcv_lookup(pyr, coords, sigma, c["S"], factor), g["pcv/%s/lookup" % name])


@pytest.mark.parametrize("name", list(_cases.NORMCORR_CASES))
def test_norm_correlation_oracles(name, golden, c_oracle):
    """CGI normalised volumes (meta_arch/cgi/submodule.py:143-180)."""
    c = _cases.NORMCORR_CASES[name]
    a, b = _cases.volume_inputs(c)
    g = golden("pcv_cgi")
    want = g["normcorr/%s/gwc_norm" % name]
    with torch.no_grad():
        assert maxabs(to.gwc_volume_norm(T(a), T(b), c["D"], c["G"]).numpy(), want) <= 1e-6
    assert maxabs(c_oracle.gwc_volume_norm(a, b, c["D"], c["G"]), want) <= 1e-6
    if c["G"] == 1:
        w1 = g["normcorr/%s/norm_corr" % name]
        with torch.no_grad():
            assert maxabs(to.norm_correlation_volume(T(a), T(b), c["D"]).numpy(), w1) <= 1e-6
        assert maxabs(c_oracle.gwc_volume_norm(a, b, c["D"], 1), w1) <= 1e-6


@pytest.mark.parametrize("name", list(_cases.CONCAT_CASES))
def test_concat_oracles(name, golden, c_oracle):
    c = _cases.CONCAT_CASES[name]
    a, b = _cases.volume_inputs(c)
    g = golden("volumes")
    for masked, key in ((True, "concat_gwcnet/"), (False, "concat_igev/")):
        with torch.no_grad():
            assert np.array_equal(to.concat_volume(T(a), T(b), c["D"], masked).numpy(), g[key + name])
        assert np.array_equal(c_oracle.concat_volume(a, b, c["D"], int(masked)), g[key + name])


@pytest.mark.parametrize("name", list(_cases.GRU_CASES))
def test_gru_oracles(name, golden, c_oracle):
    c = _cases.GRU_CASES[name]
    h, czrq, xs = _cases.gru_inputs(c)
    hd = c["hidden"]
    cin = hd + sum(c["inputs"])
    shapes = {}
    for n in ("convz", "convr", "convq"):
        shapes["g.%s.weight" % n] = (hd, cin, 3, 3)
        shapes["g.%s.bias" % n] = (hd,)
    sd = _synth.torch_state_dict(shapes, c["seed"])
    cz, cr, cq = T(czrq).split(hd, dim=1)
    want = golden("gru")[name + "/h"]
    with torch.no_grad():
        got = to.conv_gru(sd, "g", T(h), cz, cr, cq, *[T(x) for x in xs]).numpy()
    assert maxabs(got, want) <= 2e-6
    # C oracle: direct convolution + gate arithmetic
    hx = np.concatenate([h] + xs, axis=1)
    w = {k: v.numpy() for k, v in sd.items()}
    az = c_oracle.conv2d_same(hx, w["g.convz.weight"], w["g.convz.bias"])
    ar = c_oracle.conv2d_same(hx, w["g.convr.weight"], w["g.convr.bias"])
    z, rh = c_oracle.gru_gate_zr(az, ar, cz.numpy(), cr.numpy(), h)
    aq = c_oracle.conv2d_same(np.concatenate([rh] + xs, axis=1), w["g.convq.weight"], w["g.convq.bias"])
    assert maxabs(c_oracle.gru_gate_out(aq, cq.numpy(), z, h), want) <= 2e-5


@pytest.mark.parametrize("name", list(_cases.UPDATE_CASES))
def test_update_block_oracle(name, golden):
    c = _cases.UPDATE_CASES[name]
    cfg = _cases.update_cfg(c)
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "MANIFEST.json")))["state_dict_keys"][name]
    from test_layout import update_block_shapes
    shapes = update_block_shapes(c["igev"], cfg)
    assert sorted(shapes) == keys
    sd = _synth.torch_state_dict({"update_block." + k: v for k, v in shapes.items()}, c["seed"])
    net, inp, corr, flow = _cases.update_inputs(c)
    n = c["n"]
    g = golden("update")
    with torch.no_grad():
        onet = [T(x.copy()) for x in net]
        tinp = [list(T(x).split(128, dim=1)) for x in inp]
        onet, omask, odelta = to.update_block(sd, "update_block", n, onet, tinp, T(corr), T(flow),
                                              it_coarse=(n == 3), it_mid=(n >= 2), igev=c["igev"])
    for i in range(3):
        assert maxabs(onet[i].numpy(), g["%s/net%d" % (name, i)]) <= 5e-6
    assert maxabs(omask.numpy()[:, :, ::2, ::2], g[name + "/mask"]) <= 2e-5
    assert maxabs(odelta.numpy(), g[name + "/delta"]) <= 2e-5


@pytest.mark.parametrize("name", ["64x128_it4", "64x128_it12", "256x512_it8"])
def test_raft_e2e_oracle(name, golden):
    """BASELINE.json configs[0] (256x512, 8 iters, CPU) is the third case."""
    c = _cases.E2E_CASES[name]
    manifest = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "MANIFEST.json")))
    shapes = {k: tuple(v) for k, v in manifest["raft_state_dict"].items()}
    sd = _synth.torch_state_dict(shapes, _cases.E2E_WEIGHT_SEED)
    from dkt_stereo_amd.raft_stereo import BASE_CONFIG
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    lo, up = to.raft_stereo_forward(sd, BASE_CONFIG, T(i1), T(i2), c["iters"])
    g = golden("raft_e2e")
    s = int(g[name + "/stride"])
    # the tolerance north_star states for the final disparity map
    assert maxabs(up.numpy()[:, :, ::s, ::s], g[name + "/flow_up"]) <= 1e-3
    assert maxabs(lo.numpy()[:, :1], g[name + "/flow_lo"]) <= 1e-3


def test_igev_loop_oracle(golden):
    c = _cases.IGEV_LOOP_CASES["small"]
    s = c["seed"]
    cfg = dict(corr_levels=2, corr_radius=4, n_downsample=2, n_gru_layers=3,
               hidden_dims=[128, 128, 128], slow_fast_gru=False)
    from test_layout import update_block_shapes
    shapes = update_block_shapes(True, cfg)
    sd = _synth.torch_state_dict({"update_block." + k: v for k, v in shapes.items()}, s)
    m1, m2, geo, disp, coords = _cases.geo_inputs(dict(c, L=2, r=4))
    B, H, W = c["B"], c["H"], c["W"]
    net = [np.tanh(_synth.normal((B, 128, H >> i, W >> i), s, "net%d" % i)) for i in range(3)]
    inp = [_synth.normal((B, 384, H >> i, W >> i), s, "inp%d" % i, scale=0.5) for i in range(3)]
    tinp = [list(T(x).split(128, dim=1)) for x in inp]
    d, m = to.igev_iterations(sd, cfg, T(m1), T(m2), T(geo), T(np.abs(disp)), [T(x) for x in net], tinp, c["iters"])
    g = golden("igev_loop")
    assert maxabs(d.numpy(), g["small/disp"]) <= 1e-4
    assert maxabs(m.numpy(), g["small/mask"]) <= 1e-4
