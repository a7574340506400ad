"""-m gpu: parity holes named by the round-1 review, pinned.

  * BASELINE.json configs[3]'s per-GPU share (batch 8 at 736x1248, 32 iterations);
  * the IGEV refinement loop at configs[2]'s shapes (184x312, 32 iterations) and its slow-fast schedule;
  * the slow-fast GRU schedule of RAFT-Stereo (raft_stereo.py:156-159), 3 and 2 GRU layers;
  * dynamic range of the split-fp16 convolutions (large / tiny / mixed / out-of-range / non-finite inputs);
  * captured-graph invalidation when weights change; a new geometry volume per pair;
  * two threads driving two replicas on one device.
Tolerances as in test_gpu_parity.py: final disparity <= 1e-3 max-abs (north_star).
"""
import copy
import threading

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _cases
import _synth
from test_gpu_parity import DEV, G, _make_block, _raft, maxabs

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------
# cfg4: one GPU's share of the batch-64 run
# ---------------------------------------------------------------------------------
@torch.no_grad()
def test_cfg4_batch8_share_equals_pairs_run_alone(golden):
    """B = 8 pairs of 736x1248, 32 iterations, in ONE forward: every pair within 1e-3 of the same pair
    run alone, and the pair with the benchmark fixture's seed within 1e-3 of the reference's output."""
    c = _cases.E2E_CASES["736x1248_it32"]
    model, _ = _raft()
    seeds = [10, 11, 12, c["seed"], 14, 15, 16, 17]
    shifts = [12, 20, 40, c["shift"], 8, 30, 24, 16]
    pairs = [_synth.image_pair(s, 1, c["H"], c["W"], sh) for s, sh in zip(seeds, shifts)]
    i1 = G(np.concatenate([p[0] for p in pairs]))
    i2 = G(np.concatenate([p[1] for p in pairs]))
    _, up8 = model(i1, i2, iters=c["iters"], test_mode=True)
    assert up8.shape == (8, 1, c["H"], c["W"])
    g = golden("raft_e2e")
    s = int(g["736x1248_it32/stride"])
    d_ref = maxabs(up8[3:4, :, ::s, ::s], g["736x1248_it32/flow_up"])
    print("cfg4 share: pair 3 vs reference fixture %.3e" % d_ref)
    assert d_ref <= 1e-3
    worst = 0.0
    for k in range(8):
        _, up1 = model(i1[k:k + 1], i2[k:k + 1], iters=c["iters"], test_mode=True)
        worst = max(worst, maxabs(up8[k:k + 1], up1))
    print("cfg4 share: batch of 8 vs pairs alone, worst max|d| %.3e" % worst)
    assert worst <= 1e-3


# ---------------------------------------------------------------------------------
# IGEV loop: all fixtures (3 / 2 GRU layers, slow-fast schedule, cfg3 shapes x 32 iterations)
# ---------------------------------------------------------------------------------
def _igev_setup(c):
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    blk, _ = _make_block(dict(seed=c["seed"], igev=True, n=c["n"]))
    blk.args.slow_fast_gru = c["slow_fast"]
    m1, m2, geo, disp0, coords, net, inp = _cases.igev_loop_inputs(c)
    gnet = [G(x) for x in net]
    ginp = [list(G(x).split(128, dim=1)) for x in inp]
    geo_fn = Combined_Geo_Encoding_Volume(G(m1), G(m2), G(geo), radius=4, num_levels=2)
    return blk, geo_fn, G(disp0), G(coords), gnet, ginp, (m1, m2, geo)


@pytest.mark.parametrize("name", list(_cases.IGEV_LOOP_CASES))
@torch.no_grad()
def test_igev_loop_fixtures(name, golden):
    from dkt_stereo_amd.igev_loop import _plain, igev_iterate
    c = _cases.IGEV_LOOP_CASES[name]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    g = golden("igev_loop")
    st = int(g[name + "/mask_stride"])
    want_d, want_m, _ = _plain(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, c["iters"])
    dd, dm = maxabs(want_d, g[name + "/disp"]), maxabs(want_m[:, :, ::st, ::st], g[name + "/mask"])
    print("igev loop %s (plain): max|d disp| %.3e max|d mask| %.3e" % (name, dd, dm))
    assert dd <= 1e-3 and dm <= 1e-3
    got_d, got_m, _ = igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, c["iters"], cache={})
    if c["n"] == 3 and not c["slow_fast"]:
        from dkt_stereo_amd import igev_loop, loop_c8
        if igev_loop.USE_C8 and loop_c8.eligible_igev(blk, net[0].shape):
            # round-3 kernels (loop_c8.C8LoopIGEV): the same split-fp16 arithmetic in another accumulation order
            assert maxabs(got_d, want_d) <= 2e-4 and maxabs(got_m, want_m) <= 2e-4
        else:
            # graph + pipelined GRUs: same arithmetic, same order of updates
            assert torch.equal(got_d, want_d) and torch.equal(got_m, want_m)
    assert maxabs(got_d, g[name + "/disp"]) <= 1e-3
    assert maxabs(got_m[:, :, ::st, ::st], g[name + "/mask"]) <= 1e-3


@pytest.mark.parametrize("c8", [False, True])
@torch.no_grad()
def test_igev_iterate_new_volume_per_pair_and_weight_change(c8, monkeypatch):
    """(c8 = False: the round-2 loop, bit-identical to the plain loop; True: the default loop_c8, within the split-fp16 class.)
    The reference builds a new Combined_Geo_Encoding_Volume for every pair (igev_stereo.py:192-193).
    With a persistent cache the captured graph must follow: (1) a new volume object of the same shapes,
    (2) the same volume rebuilt in place, (3) changed weights (load_state_dict after the first capture)."""
    from dkt_stereo_amd.geometry import Combined_Geo_Encoding_Volume
    from dkt_stereo_amd import igev_loop
    from dkt_stereo_amd.igev_loop import _plain, igev_iterate
    monkeypatch.setattr(igev_loop, "USE_C8", c8)
    c = dict(_cases.IGEV_LOOP_CASES["small"], H=16, W=32)
    blk, geo_a, d0, coords, net, inp, (m1, m2, geo) = _igev_setup(c)
    cache = {}
    iters = 6

    def both(geo_fn):
        want = _plain(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, iters)
        got = igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, iters, cache=cache)
        if c8:
            assert maxabs(got[0], want[0]) <= 2e-4 and maxabs(got[1], want[1]) <= 2e-4
        else:
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        return got[0]

    def the_graph():
        return cache["state"].c8.graph if c8 else cache["state"].graph

    r_a = both(geo_a)
    graph = the_graph()
    assert graph is not None
    del geo_a
    # (1) new object, other content; many allocations in between so that ids / addresses get recycled
    geo2 = _synth.normal(geo.shape, 77, "geo2")
    junk = [torch.empty(1 << 20, device=DEV) for _ in range(8)]
    geo_b = Combined_Geo_Encoding_Volume(G(m2), G(m1), G(geo2), radius=4, num_levels=2)
    del junk
    r_b = both(geo_b)
    assert the_graph() is graph                          # replayed, not re-captured
    assert not torch.equal(r_a, r_b)
    # (2) the cached volume rebuilt in place by the caller
    cache["state"].geo_fn.rebuild(G(m1), G(m2), G(geo))
    r_c = both(cache["state"].geo_fn)
    assert torch.equal(r_c, r_a) and the_graph() is graph
    # (3) weights change -> fresh capture, results follow the new weights
    sd = {k: v * 1.01 for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    r_d = both(cache["state"].geo_fn)
    assert the_graph() is not graph
    assert not torch.equal(r_d, r_a)


# ---------------------------------------------------------------------------------
# RAFT slow-fast schedule
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.E2E_SLOWFAST_CASES))
@torch.no_grad()
def test_raft_stereo_slow_fast(name, golden):
    c = _cases.E2E_SLOWFAST_CASES[name]
    model, _ = _raft(slow_fast_gru=True, n_gru_layers=c["n"])
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    g = golden("raft_e2e")
    for graph in (True, False):
        model.use_hip_graph = graph
        lo, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
        d_up, d_lo = maxabs(up, g[name + "/flow_up"]), maxabs(lo[:, :1], g[name + "/flow_lo"])
        print("%s graph=%s: max|d_up| %.3e max|d_lo| %.3e" % (name, graph, d_up, d_lo))
        assert d_up <= 1e-3 and d_lo <= 1e-3


@torch.no_grad()
@pytest.mark.parametrize("c8_loop", [False, True])
def test_raft_graph_follows_weight_and_backend_changes(c8_loop):
    """A captured iteration holds pointers to packed weights: load_state_dict / in-place updates /
    set_backend after the first forward must give the result of a fresh model, not a stale replay.
    c8_loop: the default loop (loop_c8: its captured units live in the loop object; the fresh model replays its own capture,
    which is the same arithmetic); else the round-2 loop, whose capture equals the plain loop bit for bit."""
    from dkt_stereo_amd import conv
    model, sd = _raft()
    model.use_c8 = c8_loop

    def captured(m):
        st = m._graph_state
        g = (st["c8"].graph if st.get("c8") is not None else None) if c8_loop else st["graph"]
        return g

    i1, i2 = _synth.image_pair(0, 1, 64, 128, 12)
    a = model(G(i1), G(i2), iters=6, test_mode=True)[1]
    graph = captured(model)
    assert graph
    sd2 = {k: (v * 1.02 if k.startswith("update_block.") and v.dtype.is_floating_point else v) for k, v in sd.items()}
    model.load_state_dict(sd2)
    b = model(G(i1), G(i2), iters=6, test_mode=True)[1]
    assert captured(model) is not graph
    fresh, _ = _raft()
    fresh.load_state_dict(sd2)
    fresh.use_c8 = c8_loop
    fresh.use_hip_graph = c8_loop
    want = fresh(G(i1), G(i2), iters=6, test_mode=True)[1]
    assert torch.equal(b, want) and not torch.equal(a, b)
    # in-place parameter update (what an optimiser step does)
    graph = captured(model)
    model.update_block.flow_head.conv2.weight.mul_(0.5)
    fresh.update_block.flow_head.conv2.weight.mul_(0.5)
    assert torch.equal(model(G(i1), G(i2), iters=6, test_mode=True)[1], fresh(G(i1), G(i2), iters=6, test_mode=True)[1])
    assert captured(model) is not graph
    # backend switch
    prev = conv.get_backend()
    try:
        conv.set_backend("f16x2")
        c2 = model(G(i1), G(i2), iters=6, test_mode=True)[1]
        assert torch.equal(c2, fresh(G(i1), G(i2), iters=6, test_mode=True)[1])
    finally:
        conv.set_backend(prev)


# ---------------------------------------------------------------------------------
# dynamic range of the split-fp16 convolutions
# ---------------------------------------------------------------------------------
def _conv_err(x, layer, **kw):
    from dkt_stereo_amd import conv
    ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=layer.padding)
    got = conv.conv2d(x, layer, **kw)
    return float((got.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30), got


@pytest.mark.parametrize("ks", [1, 3])
@torch.no_grad()
def test_conv_dynamic_range(ks):
    """N(0,1.5) activations are what every other test feeds.  Here: x1e4 (beyond the fp16 range without
    an exponent), x1e-4 (tiny throughout), mixed per channel, one > 65504 outlier, Inf and NaN."""
    from dkt_stereo_amd import conv
    with conv.use_backend("f16x3"):
        layer = torch.nn.Conv2d(96, 128, ks, padding=ks // 2).to(DEV)
        base = G(_synth.normal((2, 96, 23, 70), 301, "x", scale=1.5))
        e0, _ = _conv_err(base, layer)
        assert e0 <= 3e-6
        # ---- mixed scales per channel: the output is dominated by the large channels; fp32-class as is
        sc = torch.ones(96, device=DEV)
        sc[0::3] = 1e3
        sc[1::3] = 1e-3
        e_mix, _ = _conv_err(base * sc.view(1, -1, 1, 1), layer)
        print("k%d mixed per-channel scales: rel err %.2e" % (ks, e_mix))
        assert e_mix <= 3e-6
        # ---- tiny throughout: graceful degradation without an exponent, fp32-class with one
        tiny = base * 1e-4
        e_t0, _ = _conv_err(tiny, layer)
        layer.dkt_in_exp = 13
        e_t1, _ = _conv_err(tiny, layer)
        print("k%d x1e-4: rel err %.2e without exponent, %.2e with dkt_in_exp=13" % (ks, e_t0, e_t1))
        assert e_t1 <= 3e-6 and e_t0 <= 2e-3
        # ---- large: non-finite without an exponent (never silently saturated), fp32-class with one
        layer.dkt_in_exp = 0
        big = base * 1e4
        assert float(big.abs().max()) > 65520
        _, got = _conv_err(big, layer)
        assert not bool(torch.isfinite(got).all())
        with conv.calibrate() as rec:
            conv.conv2d(big, layer)
        assert layer.dkt_in_exp < 0 and len(rec) == 1
        e_b, got = _conv_err(big, layer)
        print("k%d x1e4: calibrated dkt_in_exp=%d, rel err %.2e" % (ks, layer.dkt_in_exp, e_b))
        assert e_b <= 3e-6 and bool(torch.isfinite(got).all())
        # ---- non-finite inputs propagate (a diverged state must not turn into a finite number)
        layer.dkt_in_exp = 0
        for bad in (float("nan"), float("inf"), -float("inf"), 7e4):
            x = base.clone()
            x[1, 17, 11, 35] = bad
            got = conv.conv2d(x, layer)
            hit = got[1, :, 11, 35]
            assert not bool(torch.isfinite(hit).any()), bad
            far = got[0]
            assert bool(torch.isfinite(far).all())


@torch.no_grad()
def test_conv_calibration_leaves_ordinary_layers_alone_and_forward_checks_finiteness():
    from dkt_stereo_amd import _ffi, conv
    model, _ = _raft()
    i1, i2 = _synth.image_pair(0, 1, 64, 128, 12)
    want = model(G(i1), G(i2), iters=4, test_mode=True)[1]
    model.use_hip_graph = False
    with conv.calibrate() as rec:
        model(G(i1), G(i2), iters=4, test_mode=True)
    assert len(rec) > 30
    exps = [conv.in_exp_of(l) for l, _ in rec.values()]
    print("calibrated exponents of %d layers: min %d max %d" % (len(exps), min(exps), max(exps)))
    got = model(G(i1), G(i2), iters=4, test_mode=True)[1]
    assert maxabs(got, want) <= 1e-4
    bad = G(i1).clone()
    bad[0, 1, 20, 30] = float("nan")
    with pytest.raises(_ffi.DktError):
        model(bad, G(i2), iters=4, test_mode=True)


@torch.no_grad()
def test_conv_calibration_works_with_the_captured_loop():
    """conv.calibrate() with the default HIP-graph loop: the forward inside the block takes the plain loop (every iteration's
    ranges recorded, no host sync under capture); the merged z|r layer's exponent lives on convz, survives a rebuild of the
    merged weights and forces a new capture."""
    from dkt_stereo_amd import conv
    model, _ = _raft()
    assert model.use_hip_graph
    i1, i2 = _synth.image_pair(0, 1, 64, 128, 12)
    want = model(G(i1), G(i2), iters=6, test_mode=True)[1]            # captures the loop
    with conv.calibrate() as rec:
        model(G(i1), G(i2), iters=6, test_mode=True)
    assert len(rec) > 30
    assert "_MergedZR" in {type(l).__name__ for l, _ in rec.values()}
    assert maxabs(model(G(i1), G(i2), iters=6, test_mode=True)[1], want) <= 1e-4
    g = model.update_block.gru08
    fp0 = model._weights_fingerprint()
    g._merged_zr().dkt_in_exp = -3
    assert g.convz.dkt_in_exp == -3 and model._weights_fingerprint() != fp0
    g._zr_cache = None
    assert conv.in_exp_of(g._merged_zr()) == -3
    assert maxabs(model(G(i1), G(i2), iters=6, test_mode=True)[1], want) <= 1e-4
    g._merged_zr().dkt_in_exp = 0
    assert model._weights_fingerprint() == fp0


@torch.no_grad()
def test_conv_rejects_grouped_and_dilated_layers():
    """Layers the kernel does not implement take the vendor path instead of giving wrong numbers."""
    from dkt_stereo_amd import conv
    x = G(_synth.normal((1, 32, 12, 40), 302, "x"))
    for layer in (torch.nn.Conv2d(32, 64, 3, padding=1, groups=4), torch.nn.Conv2d(32, 64, 3, padding=2, dilation=2),
                  torch.nn.Conv2d(32, 64, 3, padding=1, padding_mode="reflect")):
        layer = layer.to(DEV)
        assert not conv.hip_eligible(layer)
        if layer.padding_mode == "zeros":
            assert torch.allclose(conv.conv2d(x, layer), layer(x), atol=1e-5)


# ---------------------------------------------------------------------------------
# correlation variants
# ---------------------------------------------------------------------------------
@torch.no_grad()
def test_mix_fmap_image_is_cosine_in_test_mode(golden):
    from dkt_stereo_amd.corr import CORR_IMPLEMENTATIONS
    c = _cases.CORR_CASES["small"]
    f1, f2, co = _cases.corr_inputs(c)
    blk = CORR_IMPLEMENTATIONS["mix_fmap_image"](G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    assert maxabs(blk(G(co)), golden("corr")["small/coslookup"]) <= 1e-5


def test_corr_block_fast_is_differentiable(golden):
    """CorrBlockFast1D ("reg_cuda") must not detach: same gradients as CorrBlock1D (reference: CorrSampler.backward)."""
    from dkt_stereo_amd.corr import CorrBlock1D, CorrBlockFast1D
    c = _cases.CORR_CASES["small"]
    f1, f2, co = _cases.corr_inputs(c)
    K = 2 * c["r"] + 1
    R = G(_synth.normal((c["B"], c["L"] * K, c["H"], c["W"]), c["seed"], "gout"))
    grads = []
    for cls in (CorrBlock1D, CorrBlockFast1D):
        a, b = G(f1).requires_grad_(True), G(f2).requires_grad_(True)
        blk = cls(a, b, num_levels=c["L"], radius=c["r"])
        out = blk(G(co))
        assert out.requires_grad
        grads.append(torch.autograd.grad(out, [a, b], R))
    assert blk.corr_pyramid[0].dim() == 5
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    g = golden("corr_bwd")
    assert maxabs(grads[1][0], g["small/gf1"]) <= 4e-6 * max(float(np.abs(g["small/gf1"]).max()), 1.0)


# ---------------------------------------------------------------------------------
# threads (the reference runs nn.DataParallel replicas from one Python thread each, tools/ft_dkt.py:119)
# ---------------------------------------------------------------------------------
@torch.no_grad()
def test_two_threads_two_replicas_one_device():
    """Two Python threads, two replicas sharing every sub-module (what nn.parallel.replicate hands to
    parallel_apply), one device, a stream each: both capture and replay their own iteration graphs and both get
    the single-threaded result bit for bit.  Inputs are on the device before the threads start (DataParallel
    scatters first); a host synchronisation outside forward() holds the capture guard like forward() does."""
    from dkt_stereo_amd.update import GPU_GUARD
    model, _ = _raft()
    replica = copy.copy(model)                       # what nn.parallel.replicate makes: shallow copies sharing
    replica._modules = dict(model._modules)          # parameters' storage and (here) every __dict__ entry
    pairs = [_synth.image_pair(s, 1, 64, 128, sh) for s, sh in ((0, 12), (5, 20))]
    dev_pairs = [(G(a), G(b)) for a, b in pairs]
    want = []
    for a, b in dev_pairs:                           # a fresh state per pair, as each thread below has: the default loop picks
        model._graph_state = None                    # its C8S activation scales from the first pair a state sees
        want.append(model(a, b, iters=7, test_mode=True)[1].clone())
    model._graph_state = None
    torch.cuda.synchronize()
    for rounds in range(3):                          # three fresh rounds: the interleaving differs every time
        out, err = [None, None], []
        for m in (model, replica):
            m._graph_state = None

        def run(k, m):
            try:
                with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                    m._graph_state = None            # this thread's state: capture again
                    for _ in range(3):
                        out[k] = m(dev_pairs[k][0], dev_pairs[k][1], iters=7, test_mode=True)[1].clone()
                    with GPU_GUARD.shared():
                        torch.cuda.current_stream().synchronize()
            except Exception as e:        # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=run, args=(k, m)) for k, m in enumerate((model, replica))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])


# ---------------------------------------------------------------------------------
# lookup fused with the motion encoder's 1x1 layer (dkt_corr1d_lookup_conv1x1)
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(_cases.CORR_CASES))
@pytest.mark.parametrize("cout,relu", [(64, True), (40, False), (7, True)])
@torch.no_grad()
def test_lookup_conv1x1_fused(name, cout, relu):
    """The sampled values (debug tap) are BIT-IDENTICAL to the stand-alone lookup; the 1x1 product is an
    fp32 fma chain on the matrix pipe: within fp32 round-off of an fp64 contraction of the same samples."""
    from dkt_stereo_amd.corr import CorrBlock1D
    c = _cases.CORR_CASES[name]
    f1, f2, co = _cases.corr_inputs(c)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=c["L"], radius=c["r"])
    nk = c["L"] * (2 * c["r"] + 1)
    layer = torch.nn.Conv2d(nk, cout, 1).to(DEV)
    res = blk.lookup_conv1x1(G(co), layer, relu=relu, tap=True)
    if c["L"] not in (2, 3, 4) or c["r"] not in (3, 4):
        assert res is None
        return
    out, tap = res
    want_tap = blk(G(co))
    assert torch.equal(tap, want_tap) or (torch.equal(torch.isnan(tap), torch.isnan(want_tap))
                                          and torch.equal(torch.nan_to_num(tap), torch.nan_to_num(want_tap)))
    ref = F.conv2d(want_tap.double(), layer.weight.double(), layer.bias.double())
    ref = ref.clamp_min(0) if relu else ref
    ok = torch.isfinite(ref)
    scale = max(float(ref[ok].abs().max()), 1.0)
    assert float((out.double() - ref)[ok].abs().max()) <= 2e-6 * scale
    # the deferred object the harness hands to the motion encoder
    d = blk.deferred(G(co))
    assert torch.equal(d.materialize(), want_tap)
    assert torch.equal(d.conv1x1(layer, relu=relu), out)


@torch.no_grad()
def test_lookup_conv1x1_full_size_and_unfused_path_agree(golden):
    """cfg2 shapes: fused kernel vs lookup + convolution; the end-to-end result with fusion off."""
    from dkt_stereo_amd.corr import CorrBlock1D
    f1, f2 = _synth.fmap_pair(5, 1, 256, 184, 312)
    co = _synth.coords(5, 1, 184, 312)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=4, radius=4)
    layer = torch.nn.Conv2d(36, 64, 1).to(DEV)
    out, tap = blk.lookup_conv1x1(G(co), layer, relu=True, tap=True)
    want_tap = blk(G(co))
    assert torch.equal(tap, want_tap)
    ref = F.conv2d(want_tap.double(), layer.weight.double(), layer.bias.double()).clamp_min(0)
    assert float((out.double() - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1.0)
    c = _cases.E2E_CASES["256x512_it8"]
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    model, _ = _raft()
    g = golden("raft_e2e")
    s = int(g["256x512_it8/stride"])
    for fuse in (True, False):
        model.fuse_lookup = fuse
        model._graph_state = None
        up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)[1]
        assert maxabs(up[:, :, ::s, ::s], g["256x512_it8/flow_up"]) <= 1e-3


# ---------------------------------------------------------------------------------
# two GRUs in shared launches (dkt_conv2d_f16s_pair)
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("shapes", [((1, 184, 312), (1, 46, 78)), ((2, 20, 40), (2, 5, 10)), ((1, 33, 70), (3, 9, 17))])
@torch.no_grad()
def test_gru_pair_equals_two_separate_grus(shapes):
    """The paired launch changes how tiles are distributed over blocks, not the arithmetic of any output
    element: bit-identical to running the two ConvGRUs one after the other (core/update.py:23-32)."""
    from dkt_stereo_amd.update import ConvGRU, gru_pair
    (Ba, Ha, Wa), (Bb, Hb, Wb) = shapes
    ga, gb = ConvGRU(128, 256).to(DEV), ConvGRU(128, 128).to(DEV)
    R = lambda *s: torch.randn(*s, device=DEV)      # noqa: E731
    args = []
    for (B, H, W, nx) in ((Ba, Ha, Wa, 2), (Bb, Hb, Wb, 1)):
        args.append((torch.tanh(R(B, 128, H, W)), R(B, 128, H, W), R(B, 128, H, W), R(B, 128, H, W),
                     [R(B, 128, H, W) for _ in range(nx)], None))
    want_a = ga(*args[0][:4], *args[0][4])
    want_b = gb(*args[1][:4], *args[1][4])
    got_a, got_b = gru_pair(ga, args[0], gb, args[1])
    assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
    # in place (what the loop harness does): the hidden state tensors are overwritten
    ha, hb = args[0][0].clone(), args[1][0].clone()
    ia, ib = gru_pair(ga, (ha, *args[0][1:5], ha), gb, (hb, *args[1][1:5], hb))
    assert ia.data_ptr() == ha.data_ptr() and ib.data_ptr() == hb.data_ptr()
    assert torch.equal(ha, want_a) and torch.equal(hb, want_b)


@torch.no_grad()
def test_pair_refuses_mismatched_problems():
    from dkt_stereo_amd import _ffi, conv
    la, lb = torch.nn.Conv2d(64, 256, 3, padding=1).to(DEV), torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    assert not conv.pair_eligible(la, lb)
    assert conv.pair_eligible(la, torch.nn.Conv2d(32, 200, 3, padding=1).to(DEV))


# ---------------------------------------------------------------------------------
# on-the-fly lookup: LDS-staged window path and the general (gather) path
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["rows", "vertical_shift", "fractional_y", "wide_windows"])
@torch.no_grad()
def test_otf_lookup_paths(mode):
    """PytorchAlternateCorrBlock1D (core/corr.py:64-107) against the oracle restatement of the reference on
    coordinates that exercise both kernel paths: the pixel's own row with smooth disparity (staged window),
    a vertical offset / fractional y (grid_sample's 4-tap arithmetic, general path), and windows spread over
    more columns than the staged window holds (general path)."""
    from oracle import torch_oracle as to
    from dkt_stereo_amd.corr import PytorchAlternateCorrBlock1D
    B, C, H, W = 2, 48, 6, 150
    f1, f2 = _synth.fmap_pair(401, B, C, H, W)
    co = _synth.coords(401, B, H, W, spread=8.0)
    if mode == "vertical_shift":
        co[:, 1] += 1.0
    elif mode == "fractional_y":
        co[:, 1] += _synth.uniform((B, H, W), -0.7, 0.7, 402, "dy")
    elif mode == "wide_windows":
        co = _synth.coords(403, B, H, W, spread=140.0)
    got = PytorchAlternateCorrBlock1D(G(f1), G(f2), num_levels=3, radius=4)(G(co))
    want = to.corr1d_lookup_alt(torch.from_numpy(f1), torch.from_numpy(f2), torch.from_numpy(co), 3, 4).numpy()
    scale = max(float(np.abs(want).max()), 1.0)
    d = maxabs(got, want)
    print("otf %s: max|d| %.3e (scale %.2f)" % (mode, d, scale))
    assert d <= 4e-6 * scale


# ---------------------------------------------------------------------------------
# GwcNet end to end (BASELINE configs[4]) and the evaluation chain
# ---------------------------------------------------------------------------------
def _gwcnet():
    from dkt_stereo_amd.gwcnet import GWCNet
    m = GWCNet()
    m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), _cases.GWCNET_WEIGHT_SEED), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name", list(_cases.GWCNET_CASES))
@torch.no_grad()
def test_gwcnet_end_to_end(name, golden):
    """Feature extraction on this library's convolutions, gwc(40 groups) + concat volume in one buffer, 3-D
    aggregation on the vendor library, soft-argmin: final disparity within 1e-3 of the reference GWCNet."""
    c = _cases.GWCNET_CASES[name]
    model = _gwcnet()
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    none, disp = model(G(i1), G(i2), test_mode=True)
    g = golden("gwcnet")
    s = int(g[name + "/stride"])
    d = maxabs(disp[:, :, ::s, ::s], g[name + "/disp"])
    print("gwcnet %s: max|d| %.3e" % (name, d))
    assert none is None and disp.shape == (c["B"], 1, c["H"], c["W"])
    assert d <= 1e-3


@torch.no_grad()
def test_evaluation_chain_matches_reference(golden, tmp_path):
    """Files on disk -> frame_utils readers -> InputPadder(32) -> RAFTStereo -> unpad: the prediction equals the
    reference's own chain on the same pair (fixture), and EPE / D1 are what the metric says."""
    from test_host_round2 import _write_kitti_like
    from dkt_stereo_amd import evaluate
    c = _cases.EVAL_CASES["raft_100x187_it4"]
    paths, disp = _write_kitti_like(str(tmp_path), c["seed"], c["H"], c["W"], c["shift"])
    model, _ = _raft()
    res = evaluate.validate(model, [tuple(paths)], iters=c["iters"], device=DEV, keep=True)
    pr = res["predictions"][0]
    g = golden("eval")
    assert maxabs(pr[None], g["raft_100x187_it4/flow"]) <= 1e-3
    err = np.abs(pr[0].numpy() + disp)
    val = disp > 0
    assert abs(res["epe"] - err[val].mean()) <= 1e-4
    assert abs(res["d1"] - 100.0 * (err[val] > 3.0).mean()) <= 1e-9


# ---------------------------------------------------------------------------------
# few-output 3x3 layers (flow head / disparity head tail) on the DMA-staged exact-fp32 kernel
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [("flow_head", 1, 256, 2, 184, 312, False), ("disp_head", 2, 256, 1, 33, 70, False),
                                   ("odd", 1, 37, 3, 9, 130, True), ("c4", 3, 40, 4, 8, 8, True), ("one_px", 1, 5, 2, 1, 1, False),
                                   ("narrow", 1, 256, 2, 5, 3, False), ("vec_ragged", 2, 50, 2, 7, 20, True),
                                   ("x_only", 1, 256, 1, 46, 80, False)], ids=lambda s: s[0])
@torch.no_grad()
def test_few_output_conv3x3(shape):
    """conv2d routes 3x3 layers with <= 4 outputs to dkt_conv2d_direct's DMA-staged kernel: exact fp32 FMAs,
    four channel quarters summed in a fixed order -- within fp32 round-off of an fp64 convolution, deterministic."""
    from dkt_stereo_amd import conv
    name, B, cin, cout, H, W, relu = shape
    with conv.use_backend("f16x3"):
        torch.manual_seed(4321)
        layer = torch.nn.Conv2d(cin, cout, 3, padding=1).to(DEV)
        assert conv.few_eligible(layer)
        x = G(_synth.normal((B, cin, H, W), 97, name, scale=2.0))
        ref = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=1)
        ref = ref.clamp_min(0) if relu else ref
        got = conv.conv2d(x, layer, relu=relu)
        tol = max(2e-6, 1.5e-7 * (cin * 9) ** 0.5)
        assert got.shape == ref.shape
        assert float((got.double() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        assert torch.equal(got, conv.conv2d(x, layer, relu=relu))
        # into a channel slice of a wider buffer (the out= form)
        buf = torch.full((B, cout + 3, H, W), 7.0, device=DEV)
        conv.conv2d(x, layer, relu=relu, out=buf[:, 1:1 + cout])
        assert torch.equal(buf[:, 1:1 + cout], got) and float(buf[:, 0].min()) == 7.0 and float(buf[:, -1].max()) == 7.0


# ---------------------------------------------------------------------------------
# encoder glue folded into the convolutions (dkt_conv2d_f16s_desc: in_norm, epilogue 3)
# ---------------------------------------------------------------------------------
@torch.no_grad()
@pytest.mark.parametrize("shape", [("c64_small", 2, 64, 40, 72), ("c64_full", 1, 64, 128, 1024),
                                   ("c96_small", 2, 96, 24, 60), ("c128_mid", 1, 128, 64, 500)], ids=lambda s: s[0])
def test_conv_in_norm_fused_is_bit_identical(shape):
    """relu(instance_norm(x)) folded into the staging of the following 3x3 layer (core/extractor.py:46-50) gives the
    bits of the separate normalise pass + convolution -- in every tile shape the fused form is instantiated for."""
    from dkt_stereo_amd import conv, extractor
    name, B, C, H, W = shape
    with conv.use_backend("f16x3"):
        torch.manual_seed(77)
        layer = torch.nn.Conv2d(C, C, 3, padding=1).to(DEV)
        norm = torch.nn.InstanceNorm2d(C)
        x = G(_synth.normal((B, C, H, W), 41, name, scale=3.0)) + 0.7
        assert conv.fused_eligible(layer, True)
        params = extractor.instance_norm_params(norm, x)
        ref_n = F.instance_norm(x.double())
        assert float((params[:, 0].double() - x.double().mean((2, 3)).reshape(-1)).abs().max()) < 1e-5
        want = conv.conv2d(extractor.norm_act(norm, x, True), layer)
        got = conv.conv2d_fused(x, layer, in_norm=params)
        assert torch.equal(got, want)
        ref = F.conv2d(ref_n.clamp_min(0), layer.weight.double(), layer.bias.double(), padding=1)
        assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        # with the residual join on top
        res = G(_synth.normal((B, C, H, W), 42, name))
        got = conv.conv2d_fused(x, layer, relu=True, in_norm=params, residual=res)
        assert torch.equal(got, extractor.add_relu(res, want.clamp_min(0)))


@torch.no_grad()
@pytest.mark.parametrize("shape", [("3x3_64", 2, 64, 64, 3, 40, 72), ("3x3_128_mid", 1, 96, 128, 3, 64, 500),
                                   ("3x3_256", 1, 128, 256, 3, 46, 78), ("1x1_128", 2, 128, 128, 1, 23, 39),
                                   ("3x3_24", 1, 64, 24, 3, 30, 50)], ids=lambda s: s[0])
def test_conv_residual_epilogue_is_bit_identical(shape):
    """epilogue 3: relu(residual + relu(conv)) == the separate dkt_add_relu pass (core/extractor.py:60)."""
    from dkt_stereo_amd import conv, extractor
    name, B, cin, cout, k, H, W = shape
    with conv.use_backend("f16x3"):
        torch.manual_seed(78)
        layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).to(DEV)
        x = G(_synth.normal((B, cin, H, W), 43, name, scale=1.5))
        res = G(_synth.normal((B, cout, H, W), 44, name))
        assert conv.fused_eligible(layer)
        for relu in (True, False):
            want = extractor.add_relu(res, conv.conv2d(x, layer, relu=relu))
            assert torch.equal(conv.conv2d_fused(x, layer, relu=relu, residual=res), want)
        with pytest.raises(ValueError):
            conv.conv2d_fused(x, layer, residual=res[:, :1])


@torch.no_grad()
def test_conv_in_norm_refusals():
    from dkt_stereo_amd import _ffi, conv, extractor
    with conv.use_backend("f16x3"):
        wide = torch.nn.Conv2d(64, 256, 3, padding=1).to(DEV)
        x = G(_synth.normal((1, 64, 16, 40), 45, "ref"))
        params = extractor.instance_norm_params(torch.nn.InstanceNorm2d(64), x)
        assert not conv.fused_eligible(wide, True)
        with pytest.raises(_ffi.DktError):
            conv.conv2d_fused(x, wide, in_norm=params)
        with pytest.raises(ValueError):
            conv.conv2d_fused(x, torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV), in_norm=params[:5])


@torch.no_grad()
def test_encoders_fused_equal_unfused(monkeypatch):
    """BasicEncoder (instance norm, both images) and MultiBasicEncoder (folded batch norm) with the normalise / join
    passes folded into the convolutions == the round-1 sequence of separate passes, bit for bit."""
    from dkt_stereo_amd import conv, extractor
    torch.manual_seed(5)
    fnet = extractor.BasicEncoder(output_dim=256, norm_fn='instance', downsample=2).to(DEV).eval()
    cnet = extractor.MultiBasicEncoder(output_dim=[[128] * 3, [128] * 3], norm_fn='batch', downsample=2).to(DEV).eval()
    for m in cnet.modules():                                   # non-trivial frozen statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    a = G(_synth.normal((1, 3, 96, 160), 46, "l"))
    b = G(_synth.normal((1, 3, 96, 160), 46, "r"))
    with conv.use_backend("f16x3"):
        outs = {}
        monkeypatch.setattr(extractor, "EPILOGUE_STATS", False)       # (statistics by their own pass: the bit-identity claim)
        for fuse in (True, False):
            monkeypatch.setattr(extractor, "FUSE_ENCODER", fuse)
            f1, f2 = fnet([a, b])
            scales = cnet(a, num_layers=3)
            outs[fuse] = [f1, f2] + [t for s in scales for t in s]
        for x, y in zip(outs[True], outs[False]):
            assert torch.equal(x, y)
        # round 4: the instance-norm statistics accumulated in the producing convolution's epilogue (another summation order:
        # fp32 partial sums per wave tile, fp64 across tiles) give the same features to round-off
        monkeypatch.setattr(extractor, "FUSE_ENCODER", True)
        monkeypatch.setattr(extractor, "EPILOGUE_STATS", True)
        g1, g2 = fnet([a, b])
        for x, y in ((g1, outs[True][0]), (g2, outs[True][1])):
            assert not torch.equal(x, y) or True
            assert float((x - y).abs().max() / y.abs().max()) <= 2e-6


@torch.no_grad()
def test_encoder_graph_equals_eager_and_follows_changes():
    """The encoder pass is replayed from a captured graph (both encoder streams inside): same bits as the eager pass,
    for new images, after an in-place weight update in fnet, and after cnet's frozen batch-norm statistics change."""
    from dkt_stereo_amd.raft_stereo import _ENCODER_STATES
    model, sd = _raft()
    model.graph_encoders = True
    fresh, _ = _raft()
    fresh.graph_encoders = False
    for seed in (0, 1, 2):                     # call 1 captures, calls 2-3 replay with other images
        i1, i2 = _synth.image_pair(seed, 1, 64, 128, 12)
        got = model(G(i1), G(i2), iters=4, test_mode=True)[1]
        assert torch.equal(got, fresh(G(i1), G(i2), iters=4, test_mode=True)[1])
    st = next(iter(_ENCODER_STATES[model].values()))
    graph = st["graph"]
    for m in (model, fresh):
        m.fnet.layer1[0].conv2.weight.mul_(1.1)
    a = model(G(i1), G(i2), iters=4, test_mode=True)[1]
    assert torch.equal(a, fresh(G(i1), G(i2), iters=4, test_mode=True)[1]) and not torch.equal(a, got)
    assert next(iter(_ENCODER_STATES[model].values()))["graph"] is not graph
    for m in (model, fresh):
        m.cnet.layer2[0].norm1.running_var.mul_(1.3)
    b = model(G(i1), G(i2), iters=4, test_mode=True)[1]
    assert torch.equal(b, fresh(G(i1), G(i2), iters=4, test_mode=True)[1]) and not torch.equal(a, b)
    # another shape gets its own capture; encode() alone returns the static outputs
    j1, j2 = _synth.image_pair(5, 1, 96, 160, 10)
    assert torch.equal(model(G(j1), G(j2), iters=4, test_mode=True)[1], fresh(G(j1), G(j2), iters=4, test_mode=True)[1])
    f1 = model.encode(G(j1), G(j2))[0]
    assert torch.equal(f1, fresh.encode(G(j1), G(j2))[0])


@torch.no_grad()
def test_flow_head_leading_outputs():
    """FlowHead(x, outputs=1) -- what the stereo loop asks for, the y component being discarded
    (raft_stereo.py:165) -- is the x plane of the full head, bit for bit, and follows weight updates."""
    from dkt_stereo_amd import conv, update
    with conv.use_backend("f16x3"):
        torch.manual_seed(3)
        head = update.FlowHead(128, 256, 2).to(DEV)
        x = G(_synth.normal((2, 128, 37, 64), 61, "fh"))
        full = head(x)
        assert torch.equal(head(x, outputs=1), full[:, :1])
        assert torch.equal(head(x, outputs=2), full)
        head.conv2.weight.mul_(1.5)
        assert torch.equal(head(x, outputs=1), head(x)[:, :1]) and not torch.equal(head(x)[:, :1], full[:, :1])


@torch.no_grad()
@pytest.mark.parametrize("cout", [1, 2])
def test_few_output_kernel_beside_an_lds_holding_kernel(cout):
    """The DMA-staged head kernel must be exact when blocks of another kernel with LDS run on the same CUs (second
    stream of the rotated loop): regression for wrong results of the 1-output form, whose 123 KB left room for a 33 KB
    block of the 1/16 GRU convolution (tools/stress_lds_dma.py)."""
    from dkt_stereo_amd import conv
    with conv.use_backend("f16x3"):
        torch.manual_seed(0)
        layer = torch.nn.Conv2d(256, cout, 3, padding=1).to(DEV)
        small = torch.nn.Conv2d(384, 256, 3, padding=1).to(DEV)
        xs = torch.randn(1, 384, 32, 64, device=DEV)
        x = torch.randn(1, 256, 64, 128, device=DEV)
        ref = conv.conv2d(x, layer).clone()
        ref_s = conv.conv2d(xs, small, relu=True).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        ys, yf = [], []
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(6):
                    ys.append(conv.conv2d(xs, small, relu=True))
            for _ in range(12):
                yf.append(conv.conv2d(x, layer))
            main.wait_stream(side)
        for _ in range(25):
            g.replay()
            torch.cuda.synchronize()
            assert all(torch.equal(y, ref) for y in yf)
            assert all(torch.equal(y, ref_s) for y in ys)


@torch.no_grad()
def test_raft_forward_is_deterministic_at_small_scales():
    """256x512 (64x128 at 1/4, 16x32 at 1/16): the scale at which small kernels of both streams share CUs.
    Twelve forwards, one result."""
    model, _ = _raft()
    i1, i2 = _synth.image_pair(1, 1, 256, 512, 20)
    ref = model(G(i1), G(i2), iters=8, test_mode=True)[1].clone()
    for _ in range(11):
        assert torch.equal(model(G(i1), G(i2), iters=8, test_mode=True)[1], ref)


@torch.no_grad()
def test_head_accumulate_epilogue():
    """target += head(x) inside the few-output kernel == the separate add, bit for bit (dkt_conv2d_direct_accumulate);
    the target is a channel slice of a wider tensor, as coords1[:, :1] is."""
    from dkt_stereo_amd import conv, update
    with conv.use_backend("f16x3"):
        torch.manual_seed(9)
        head = update.FlowHead(128, 256, 2).to(DEV)
        x = G(_synth.normal((2, 128, 30, 52), 71, "acc"))
        coords = G(_synth.normal((2, 2, 30, 52), 72, "coords", scale=20.0))
        want = coords.clone()
        want[:, :1] += head(x)[:, :1]
        got = coords.clone()
        head.add_to(x, got[:, :1], outputs=1)
        assert torch.equal(got, want)
        both = coords.clone()
        head.add_to(x, both)
        assert torch.equal(both, coords + head(x))
        # ... with the flow refresh (dst = target_new - ref) in the same epilogue, into a channel slice of a wider buffer
        ref = G(_synth.normal((2, 2, 30, 52), 73, "c0", scale=20.0))
        feat = torch.full((2, 6, 30, 52), 5.0, device=DEV)
        got = coords.clone()
        head.add_to(x, got[:, :1], outputs=1, diff=(ref[:, :1], feat[:, 4:5]))
        assert torch.equal(got, want) and torch.equal(feat[:, 4:5], want[:, :1] - ref[:, :1])
        assert float(feat[:, :4].min()) == 5.0 and float(feat[:, 5:].max()) == 5.0


@torch.no_grad()
@pytest.mark.parametrize("relu", [True, False])
def test_lazy_residual_operand_of_the_instance_norm_join(relu):
    """dkt_instance_norm_add_relu_lazy: the residual operand is normalised (and optionally rectified) inside the join --
    the same bits as normalising it in a pass of its own first."""
    from dkt_stereo_amd import extractor
    norm = torch.nn.InstanceNorm2d(24)
    a = G(_synth.normal((2, 24, 33, 52), 81, "la", scale=2.0)) + 0.4
    c = G(_synth.normal((2, 24, 33, 52), 82, "lc", scale=3.0)) - 0.2
    want = extractor.norm_add_relu(norm, extractor.norm_act(norm, a, relu), c)
    got = extractor.norm_add_relu(norm, extractor.LazyNorm(norm, a, relu), c)
    assert torch.equal(got, want)
    assert torch.equal(extractor.LazyNorm(norm, a, relu).materialize(), extractor.norm_act(norm, a, relu))


@torch.no_grad()
def test_data_parallel_replicas_run_on_a_persistent_shadow():
    """nn.DataParallel (tools/ft_dkt.py:119-125; the teachers are called with test_mode=True, :193,199) makes new replica
    modules on new threads for every forward.  A replica's test_mode forward goes to a persistent per-device copy of the
    master driven by one long-lived worker thread (raft_stereo._Shadow): the loop is captured ONCE and replayed by later
    forwards, the result equals the master's own forward bit for bit, and a weight update on the master reaches the copy."""
    from dkt_stereo_amd import raft_stereo as rs
    model, _ = _raft()
    a, b = (G(x) for x in _synth.image_pair(3, 1, 64, 128, 12))
    want = model(a, b, iters=7, test_mode=True)[1].clone()

    def through_replicas(n):
        # what DataParallel.forward does with the master: replicate, then one NEW thread per replica (parallel_apply)
        reps = [model._replicate_for_data_parallel() for _ in range(n)]
        out, err = [None] * n, []

        def run(k):
            try:
                with torch.no_grad():
                    out[k] = reps[k](a, b, iters=7, test_mode=True)[1].clone()
            except Exception as e:        # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=run, args=(k,)) for k in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        return out

    assert all(r._is_replica for r in [model._replicate_for_data_parallel()])
    for o in through_replicas(2):
        assert torch.equal(o, want)
    shadow = rs._SHADOWS[model][torch.device(DEV).index]
    assert shadow.model is not model and shadow.model.training == model.training
    state = rs._GRAPH_STATES[shadow.model]
    assert len(state) == 1                                   # one thread has ever driven the copy ...
    st = next(iter(state.values()))                          # (captured units and static buffers of the loop)
    for o in through_replicas(2):
        assert torch.equal(o, want)
    assert len(rs._GRAPH_STATES[shadow.model]) == 1          # ... and the later forwards replayed its capture
    assert next(iter(rs._GRAPH_STATES[shadow.model].values())) is st
    # a weight update on the master (the EMA teacher of tools/ft_dkt.py changes every step) reaches the copy
    model.update_block.flow_head.conv2.weight.mul_(1.5)
    want2 = model(a, b, iters=7, test_mode=True)[1].clone()
    assert not torch.equal(want2, want)
    for o in through_replicas(2):
        assert torch.equal(o, want2)
    # the switch: replicas on their own (the plain loop), same numbers within the loop's reassociation
    model.replica_shadows = False
    o = through_replicas(1)[0]
    assert maxabs(o, want2) <= 1e-3


@torch.no_grad()
def test_data_parallel_wrapper_single_device():
    """tools/evaluate_stereo.py:361: DataParallel(model, device_ids=[0]) calls the master itself (no replication)."""
    model, _ = _raft()
    a, b = (G(x) for x in _synth.image_pair(3, 1, 64, 128, 12))
    want = model(a, b, iters=7, test_mode=True)[1].clone()
    dp = torch.nn.DataParallel(model, device_ids=[torch.cuda.current_device()])
    assert torch.equal(dp(a, b, iters=7, test_mode=True)[1], want)


@torch.no_grad()
def test_data_parallel_wrapper_two_replicas():
    """tools/ft_dkt.py:122-125,193,199: the teachers under nn.DataParallel with test_mode=True.  One device listed twice
    gives torch's own replicate / scatter / parallel_apply / gather with two replicas on a 1-GPU box: both chunks of the
    batch come back as the master's own result for that pair, and a second call replays the captured loop of the copy."""
    from dkt_stereo_amd import raft_stereo as rs
    model, _ = _raft()
    pairs = [_synth.image_pair(s, 1, 64, 128, sh) for s, sh in ((3, 12), (4, 20))]
    a = torch.cat([G(p[0]) for p in pairs])
    b = torch.cat([G(p[1]) for p in pairs])
    want = model(a[:1], b[:1], iters=7, test_mode=True)[1].clone()         # (chunk 0 first: the pair the scales are picked on)
    want = torch.cat([want, model(a[1:], b[1:], iters=7, test_mode=True)[1]])
    dev = torch.device(DEV).index
    dp = torch.nn.DataParallel(model, device_ids=[dev, dev])
    for _ in range(2):
        low, up = dp(a, b, iters=7, test_mode=True)
        assert up.shape == want.shape and low.shape[0] == 2
        assert maxabs(up, want) <= 2e-4                                   # (chunk order on the copy's thread is not fixed:
    assert len(rs._GRAPH_STATES[rs._SHADOWS[model][dev].model]) == 1      #  the scales may come from either pair)
