"""-m gpu: round 5.

* tests that pin the PATH, not only the numbers: at the benchmark workload the loop must be the fused
  ConvGRU launch + the fused motion front with its error word clear and 9 dispatches per captured unit; the same at batch 8
  (BASELINE cfg4's per-GPU share) and for IGEV's loop; a declined fused launch gives the two-launch result bit for bit;
* the instance-norm statistics scratch is not overrun for heights off the tile grid;
* the non-default backbones of raft_stereo.py:43-54 against reference fixtures;
* bench.py's N > 1 branch, two ranks sharing this device over gloo.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _cases
import _synth
from test_gpu_parity import DEV, G, _raft, maxabs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

#: the dispatches of one captured RAFT unit (loop_c8.C8Loop.unit): fused ConvGRU step, head conv1 (+ folded conv2), motion front,
#: convc2 | convf2, encoder.conv on the main queue; pool | interp, gru16 z|r, gru16 q, interp | pool on the forked one
UNIT_LAUNCHES = ["dkt_gru_c8", "dkt_conv2d_c8", "dkt_motion_front_c8", "dkt_resample_pair_c8", "dkt_conv2d_c8", "dkt_conv2d_c8",
                 "dkt_resample_pair_c8", "dkt_conv2d_c8_pair", "dkt_conv2d_c8"]


def _assert_fast_path(lp, what):
    assert lp is not None, what + ": the C8S loop did not run"
    assert lp.fuse_gru, what + ": the fused ConvGRU launch was declined (two-launch form ran)"
    assert lp.front, what + ": the fused motion front did not run"
    assert not lp.take_error(), what + ": a fused ConvGRU launch timed out on a neighbour flag"
    assert lp.unit_launches is not None and sorted(lp.unit_launches) == sorted(UNIT_LAUNCHES), \
        what + ": a captured unit is %r" % (lp.unit_launches,)


@torch.no_grad()
def test_benchmark_workload_runs_the_fast_path(golden):
    """736x1248 / 32 iterations, batch 1: fused ConvGRU launch, fused front, error word clear, 9 dispatches per unit -- and the
    fixture bound through exactly that path."""
    c = _cases.E2E_CASES["736x1248_it32"]
    model, _ = _raft()
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"]))
    model(i1, i2, iters=c["iters"], test_mode=True)
    _, up = model(i1, i2, iters=c["iters"], test_mode=True)
    _assert_fast_path(model._graph_state.get("c8"), "B=1")
    g = golden("raft_e2e")
    s = int(g["736x1248_it32/stride"])
    assert maxabs(up[:, :, ::s, ::s], g["736x1248_it32/flow_up"]) <= 1e-3


@torch.no_grad()
def test_batch8_share_runs_the_fast_path():
    """BASELINE cfg4's per-GPU share (8 pairs per launch): the same path, every block several tiles."""
    model, _ = _raft()
    pairs = [_synth.image_pair(1000 + j, 1, 736, 1248, 12 if j % 2 == 0 else 40) for j in range(8)]
    i1 = torch.cat([torch.from_numpy(p[0]) for p in pairs]).to(DEV)
    i2 = torch.cat([torch.from_numpy(p[1]) for p in pairs]).to(DEV)
    model(i1, i2, iters=4, test_mode=True)
    _, up = model(i1, i2, iters=4, test_mode=True)
    assert bool(torch.isfinite(up).all())
    _assert_fast_path(model._graph_state.get("c8"), "B=8")


@torch.no_grad()
def test_igev_loop_runs_the_fused_gru_launch():
    from test_gpu_round2 import _igev_setup
    from dkt_stereo_amd import igev_loop
    c = _cases.IGEV_LOOP_CASES["kitti"]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    cache = {}
    for _ in range(2):
        igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, 4, cache=cache)
    lp = cache["state"].c8
    assert lp.fuse_gru and not lp.take_error()
    assert lp.unit_launches is not None and lp.unit_launches.count("dkt_gru_c8") == 1 and len(lp.unit_launches) <= 11, lp.unit_launches


@torch.no_grad()
def test_declined_fused_launch_equals_two_launch_form(monkeypatch):
    """dkt_gru_c8_pair answering DKT_E_UNSUPPORTED (a device that cannot hold the launch) must leave exactly the two-launch
    loop: same bits as DKT_C8_FUSE_GRU=0, and the loop object says which form ran."""
    from dkt_stereo_amd import conv_c8, loop_c8
    H, W = 544, 960                      # 136 x 240 at 1/4 resolution: 136 tiles >= FUSE_GRU_MIN_TILES
    i1, i2 = (G(t) for t in _synth.image_pair(5, 1, H, W, 12))
    monkeypatch.setattr(loop_c8, "FUSE_GRU", False)
    m0, _ = _raft()
    _, want = m0(i1, i2, iters=5, test_mode=True)
    assert not m0._graph_state["c8"].fuse_gru
    monkeypatch.setattr(loop_c8, "FUSE_GRU", True)
    monkeypatch.setattr(conv_c8, "gru_launch", lambda *a, **k: False)
    m1, _ = _raft()
    _, got = m1(i1, i2, iters=5, test_mode=True)
    lp = m1._graph_state["c8"]
    assert not lp.fuse_gru and "dkt_gru_c8" not in lp.unit_launches
    assert torch.equal(got, want)
    _, again = m1(i1, i2, iters=5, test_mode=True)
    assert torch.equal(again, want)


@torch.no_grad()
def test_flag_timeout_falls_back_and_recomputes(monkeypatch):
    """A raised error word (a fused launch that gave up on a neighbour flag) must not surface as a wrong disparity: forward
    warns, clears the word, takes the two-launch form and computes the pair again."""
    H, W = 544, 960
    i1, i2 = (G(t) for t in _synth.image_pair(6, 1, H, W, 12))
    model, _ = _raft()
    _, first = model(i1, i2, iters=4, test_mode=True)
    lp = model._graph_state["c8"]
    assert lp.fuse_gru
    lp.err.fill_(1)
    with pytest.warns(UserWarning, match="timed out"):
        _, second = model(i1, i2, iters=4, test_mode=True)
    assert not lp.fuse_gru and int(lp.err.item()) == 0
    assert maxabs(second, first) <= 2e-4                     # the other summation order of the two-launch form
    _, third = model(i1, i2, iters=4, test_mode=True)        # no warning, same path
    assert torch.equal(third, second)


@pytest.mark.parametrize("shape", [(3, 32, 40, 33, 70, 1, 3), (2, 64, 64, 37, 45, 1, 3), (1, 64, 96, 75, 90, 2, 3), (2, 48, 96, 21, 64, 1, 3),
                                   (1, 64, 96, 66, 50, 2, 1)])
def test_conv_statistics_scratch_is_not_overrun(shape):
    """dkt_conv2d_stats_ws_floats must cover the entries of waves whose rows lie past Ho (heights off the 4 / 8-row tile grid):
    the scratch sits in front of a guard area that must keep its pattern."""
    from dkt_stereo_amd import _ffi, conv
    B, cin, cout, H, W, stride, k = shape
    torch.manual_seed(H)
    layer = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).to(DEV)
    x = torch.randn(B, cin, H, W, device=DEV)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    n = int(_ffi.lib().dkt_conv2d_stats_ws_floats(B, cout, Ho, Wo))
    guard = torch.full((n + 65536,), -12345.0, device=DEV)
    with conv.use_backend("f16x3"):
        if not conv.stats_eligible(layer):
            pytest.skip("layer not on the statistics epilogue")
        conv.conv2d_stats(x, layer, _ws=guard[:n])
    torch.cuda.synchronize()
    assert bool((guard[n:] == -12345.0).all()), "the statistics epilogue wrote past dkt_conv2d_stats_ws_floats"


@pytest.mark.parametrize("name", list(_cases.E2E_BACKBONE_CASES))
@torch.no_grad()
def test_raft_backbone_variants(name, golden):
    """shared_backbone / backbone_type='interpolate' (raft_stereo.py:43-54, 97-108) against the reference's own outputs."""
    c = _cases.E2E_BACKBONE_CASES[name]
    model, _ = _raft(**c["over"])
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    lo, up = model(G(i1), G(i2), iters=c["iters"], test_mode=True)
    g = golden("raft_backbones")
    d_up, d_lo = maxabs(up, g[name + "/flow_up"]), maxabs(lo[:, :1], g[name + "/flow_lo"])
    print("%s: max|d_up| %.3e max|d_lo| %.3e" % (name, d_up, d_lo))
    assert d_up <= 1e-3 and d_lo <= 1e-3


def test_bench_two_ranks_over_gloo_on_one_device(tmp_path):
    """bench.py's world > 1 branch: two ranks (both on device 0, gloo) -- per-rank timing gather, result gather to rank 0, the
    rank != 0 early return, one JSON line from rank 0."""
    from test_dist_cpu import _free_port
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--height", "64",
                                       "--width", "128", "--iters", "4", "--steps", "2", "--warmup", "1", "--batch", "2", "--skip-cpu-baseline"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["collective_backend"] == "gloo"
    assert len(j["per_rank_pairs_per_s"]) == 2 and all(v > 0 for v in j["per_rank_pairs_per_s"])
    assert j["config"]["per_gpu_batch"] == 2 and j["gathered_batch"] == 4
    assert abs(j["value"] - 2 * 2 * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) <= 1e-6 * j["value"]


# ---- reduced-pass convolutions (dkt_conv_c8_desc.passes / dkt_gru_c8_desc.passes): the arithmetic they are DEFINED to be -----------
def _f16r(x):
    return x.half().float()


def _w_hi(w):
    """What the packed image's hi plane holds: fp16 of w * 2^e (max |w * 2^e| in [2^12, 2^13)), back at w's scale."""
    import math
    e = 12 - math.floor(math.log2(float(w.abs().max())))
    return (w * 2.0 ** e).half().float() * 2.0 ** -e


@torch.no_grad()
@pytest.mark.parametrize("passes", [1, 2])
@pytest.mark.parametrize("case", [(1, [128, 64, 64], 256, 72, 100, 1), (2, [128], 128, 50, 70, 2), (1, [64, 64], 64, 37, 45, 3),
                                  (1, [128, 128, 128], 64, 23, 39, 4), (3, [32], 64, 16, 33, 4)])
def test_conv_c8_reduced_passes_are_the_rounded_operand_convolution(case, passes):
    """passes = 2 is the fp32-class convolution of activations ROUNDED to fp16; passes = 1 additionally rounds the weights (their
    hi plane).  Against fp64 convolutions of exactly those operands the result must be in the split-fp16 class (3e-6): a
    fragment read too early or a mis-counted wait in the reduced-pass steps would not be."""
    import torch.nn.functional as F
    from dkt_stereo_amd import conv_c8 as c8
    B, chans, cout, H, W, cfg = case
    torch.manual_seed(cout + H + passes)
    layer = torch.nn.Conv2d(sum(chans), cout, 3, padding=1).to(DEV)
    xs = [torch.randn(B, c, H, W, device=DEV) for c in chans]
    with c8.passes(passes):
        got = c8.conv2d_c8([c8.pack(x) for x in xs], layer, relu=False, cfg=cfg)
    x = torch.cat([_f16r(t) for t in xs], 1).double()
    w = (_w_hi(layer.weight) if passes == 1 else layer.weight).double()
    want = F.conv2d(x, w, layer.bias.double(), padding=1)
    full = F.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), padding=1)
    rel = float((got.double() - want).abs().max() / want.abs().max())
    off = float((got.double() - full).abs().max() / full.abs().max())
    print("passes %d cfg %d: vs rounded-operand fp64 %.2e, vs the exact convolution %.2e" % (passes, cfg, rel, off))
    assert rel <= 3e-6
    assert off >= 1e-5        # (and it IS the reduced arithmetic: the full product would sit at 1e-6)


@torch.no_grad()
@pytest.mark.parametrize("passes", [1, 2])
@pytest.mark.parametrize("B,H,W,xch", [(1, 96, 160, [128, 128]), (1, 46, 78, [128]), (2, 184, 312, [128, 128])])
def test_fused_gru_reduced_passes(B, H, W, xch, passes):
    """The one-launch ConvGRU step at 1 / 2 passes against its fp64 definition on the rounded operands (r*h is rounded where it
    is stored: the q convolution reads its hi plane), and bit-reproducible launch after launch."""
    import torch.nn.functional as F
    from test_gpu_round4 import _State, _make
    from dkt_stereo_amd import conv_c8 as c8
    gru, h, xs, cz, cr, cq = _make(B, H, W, xch, seed=H + passes)
    st = _State(gru, h, xs, cz, cr, cq)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    with c8.passes(passes):
        d = st.desc()
    assert c8.gru_launch(d, err=err)
    p = {k: v.double() for k, v in gru.state_dict().items()}
    rw = (lambda w: _w_hi(w.float()).double()) if passes == 1 else (lambda w: w)
    x = torch.cat([_f16r(t) for t in xs], 1).double()
    hh = _f16r(h).double()
    hx = torch.cat([hh, x], 1)
    z = torch.sigmoid(F.conv2d(hx, rw(p["convz.weight"]), p["convz.bias"], padding=1) + cz.double())
    r = torch.sigmoid(F.conv2d(hx, rw(p["convr.weight"]), p["convr.bias"], padding=1) + cr.double())
    rh = _f16r((r.float() * h)).double()
    q = torch.tanh(F.conv2d(torch.cat([rh, x], 1), rw(p["convq.weight"]), p["convq.bias"], padding=1) + cq.double())
    want = (1 - z) * h.double() + z * q
    rel = float((st.h.double() - want).abs().max() / want.abs().max())
    print("gru passes %d: %.2e" % (passes, rel))
    # (r*h is rounded from the kernel's own fp32 r, which differs from the fp64 one in the last bits: an fp16 rounding boundary
    # crossed here and there moves single q inputs by 2^-11 relative -- hence the looser bound)
    assert rel <= 2e-4 and int(err.item()) == 0
    st2 = _State(gru, h, xs, cz, cr, cq)
    with c8.passes(passes):
        assert c8.gru_launch(st2.desc(), err=err)
    assert torch.equal(st2.h, st.h)


@torch.no_grad()
def test_precision_schedule_runs_and_default_is_untouched(golden):
    """A (k1, k2) schedule replays units of three kinds from their own captured graphs and lands near the fp32-class result; with the
    schedule removed the model is bit for bit the parity path again."""
    c = _cases.E2E_CASES["256x512_it32"]
    model, _ = _raft()
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"]))
    model.precision_schedule = None
    _, ref = model(i1, i2, iters=c["iters"], test_mode=True)
    model.precision_schedule = (12, 8)
    _, a = model(i1, i2, iters=c["iters"], test_mode=True)
    _, b = model(i1, i2, iters=c["iters"], test_mode=True)
    lp = model._graph_state["c8"]
    assert torch.equal(a, b) and {k[0] for d in (lp.graph, lp.graph_n, lp.graph_last) for k in d} == {1, 2, 3}
    d = maxabs(a, ref)
    print("schedule (12, 8) vs fp32-class: %.3e" % d)
    assert 0 < d <= 5e-2
    model.precision_schedule = None
    _, again = model(i1, i2, iters=c["iters"], test_mode=True)
    assert torch.equal(again, ref)
    g = golden("raft_e2e")
    s = int(g["256x512_it32/stride"])
    assert maxabs(again[:, :, ::s, ::s], g["256x512_it32/flow_up"]) <= 1e-3


# ---- chain launches (dkt_conv2d_c8_chain): two dependent layers, one launch, a flag round instead of the kernel boundary -----------
@torch.no_grad()
@pytest.mark.parametrize("B,H,W,blocks", [(1, 184, 312, 256), (2, 92, 156, 96), (1, 37, 45, 0), (3, 64, 128, 40)])
def test_chain_motion_tail_equals_separate_launches(B, H, W, blocks):
    """convc2 | convf2 -> encoder.conv (core/update.py:78-80,84-85) as one launch: the same bits as the pair launch + the single
    launch, launch after launch on the same flag words (every block several tiles, waits across rounds)."""
    from dkt_stereo_amd import conv_c8 as c8
    torch.manual_seed(H + B)
    c2, f2 = (torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV) for _ in range(2))
    cv = torch.nn.Conv2d(128, 126, 3, padding=1).to(DEV)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    flags = None
    for step in range(3):
        cor, flo = (c8.pack(torch.randn(B, 64, H, W, device=DEV)) for _ in range(2))
        flow = torch.randn(B, 2, H, W, device=DEV)
        cf_a, cf_b = c8.ActC8(B, 128, H, W, DEV), c8.ActC8(B, 128, H, W, DEV)
        mf_a, mf_b = c8.ActC8(B, 128, H, W, DEV, tail=2), c8.ActC8(B, 128, H, W, DEV, tail=2)
        d0 = c8.desc([cor], c2, relu=True, out_c8=cf_a, out_c8_ch0=0)
        d1 = c8.desc([flo], f2, relu=True, out_c8=cf_a, out_c8_ch0=64)
        c8.launch_pair(d0, d1, flow, 4)
        c8.conv2d_c8([cf_a], cv, relu=True, out_c8=mf_a, tail=flow, cfg=3)
        e0 = c8.desc([cor], c2, relu=True, out_c8=cf_b, out_c8_ch0=0)
        e1 = c8.desc([flo], f2, relu=True, out_c8=cf_b, out_c8_ch0=64)
        e2 = c8.desc([cf_b], cv, relu=True, out_c8=mf_b, tail=flow)
        if flags is None:
            flags = c8.chain_flags(e0, 4, 2, DEV)
        assert c8.launch_chain(e0, e1, 4, e2, 3, flags, flow, err=err, max_blocks=blocks)
        assert torch.equal(cf_b.t, cf_a.t) and torch.equal(mf_b.t, mf_a.t), step
    assert int(err.item()) == 0 and int(flags.min()) == 3 and int(flags.max()) == 3


@torch.no_grad()
@pytest.mark.parametrize("B,H,W,blocks", [(1, 92, 156, 256), (2, 46, 78, 64), (4, 92, 156, 256)])
def test_chain_gru_equals_two_launches(B, H, W, blocks):
    """The middle ConvGRU (z|r + gates -> q + in-place state update, core/update.py:23-32) as one chain launch, three dependent
    steps: bit for bit the two-launch form."""
    from test_gpu_round4 import _State, _make
    from dkt_stereo_amd import conv_c8 as c8
    args = _make(B, H, W, [128, 128], seed=7 + B)
    a, b = _State(*args), _State(*args)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    z = torch.empty_like(b.h)
    flags = None
    for step in range(3):
        za = c8.gate_zr([a.hc8, *a.xs], a.gru._merged_zr(), a.cz, a.cr, a.h, rh_c8=a.rh, cfg=4)
        c8.gate_out([a.rh, *a.xs], a.gru.convq, a.cq, za, a.h, a.h, out_c8=a.hc8, cfg=4)
        d0 = c8.desc([b.hc8, *b.xs], b.gru._merged_zr(), out=z, epilogue=1, e0=b.cz, e1=b.cr, h=b.h, out2_c8=b.rh)
        d1 = c8.desc([b.rh, *b.xs], b.gru.convq, out=b.h, out_c8=b.hc8, epilogue=2, e0=b.cq, e1=z, h=b.h)
        if flags is None:
            flags = c8.chain_flags(d0, 4, 1, DEV)
        assert c8.launch_chain(d0, None, 4, d1, 4, flags, b.h, err=err, max_blocks=blocks)
        assert torch.equal(b.h, a.h) and torch.equal(b.hc8.t, a.hc8.t) and torch.equal(b.rh.t, a.rh.t), step
    assert int(err.item()) == 0


@torch.no_grad()
def test_loop_with_chain_launches_equals_default(monkeypatch):
    """The whole forward with both chains on (7 dispatches per unit) against the default loop: the same bits."""
    from dkt_stereo_amd import loop_c8
    c = _cases.E2E_CASES["736x1248_it32"]
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"]))
    m0, _ = _raft()
    _, want = m0(i1, i2, iters=8, test_mode=True)
    monkeypatch.setattr(loop_c8, "CHAIN", 1)
    m1, _ = _raft()
    _, got = m1(i1, i2, iters=8, test_mode=True)
    _, again = m1(i1, i2, iters=8, test_mode=True)
    lp = m1._graph_state["c8"]
    assert lp.unit_launches.count("dkt_conv2d_c8_chain") == 2 and len(lp.unit_launches) == 7, lp.unit_launches
    assert not lp.take_error()
    assert torch.equal(got, want) and torch.equal(again, want)


@torch.no_grad()
@pytest.mark.parametrize("passes", [2, 1])
def test_chain_launch_at_reduced_passes_equals_separate_launches(passes):
    """The chain launch honours dkt_conv_c8_desc.passes like the single launches (the same step sequences, one kernel)."""
    from dkt_stereo_amd import conv_c8 as c8
    B, H, W = 1, 92, 156
    torch.manual_seed(passes)
    c2, f2 = (torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV) for _ in range(2))
    cv = torch.nn.Conv2d(128, 126, 3, padding=1).to(DEV)
    cor, flo = (c8.pack(torch.randn(B, 64, H, W, device=DEV)) for _ in range(2))
    flow = torch.randn(B, 2, H, W, device=DEV)
    cf_a, cf_b = c8.ActC8(B, 128, H, W, DEV), c8.ActC8(B, 128, H, W, DEV)
    mf_a, mf_b = c8.ActC8(B, 128, H, W, DEV, tail=2), c8.ActC8(B, 128, H, W, DEV, tail=2)
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    with c8.passes(passes):
        c8.launch_pair(c8.desc([cor], c2, relu=True, out_c8=cf_a, out_c8_ch0=0), c8.desc([flo], f2, relu=True, out_c8=cf_a, out_c8_ch0=64), flow, 4)
        c8.conv2d_c8([cf_a], cv, relu=True, out_c8=mf_a, tail=flow, cfg=3)
        e0 = c8.desc([cor], c2, relu=True, out_c8=cf_b, out_c8_ch0=0)
        e1 = c8.desc([flo], f2, relu=True, out_c8=cf_b, out_c8_ch0=64)
        e2 = c8.desc([cf_b], cv, relu=True, out_c8=mf_b, tail=flow)
        assert e0.passes == passes and e2.passes == passes
        assert c8.launch_chain(e0, e1, 4, e2, 3, c8.chain_flags(e0, 4, 2, DEV), flow, err=err, max_blocks=128)
    assert torch.equal(cf_b.t, cf_a.t) and torch.equal(mf_b.t, mf_a.t) and int(err.item()) == 0
    # and it is the reduced arithmetic: the fp32-class launch differs
    cf_c = c8.ActC8(B, 128, H, W, DEV)
    c8.launch_pair(c8.desc([cor], c2, relu=True, out_c8=cf_c, out_c8_ch0=0), c8.desc([flo], f2, relu=True, out_c8=cf_c, out_c8_ch0=64), flow, 4)
    assert not torch.equal(cf_c.t, cf_a.t)


@torch.no_grad()
def test_igev_loop_under_a_precision_schedule(monkeypatch):
    """IGEV's loop (C8LoopIGEV) takes the module-level schedule: units of three kinds from their own graphs, a result near the
    fp32-class one, reproducible; without the schedule the default result bit for bit."""
    from test_gpu_round2 import _igev_setup
    from dkt_stereo_amd import igev_loop, loop_c8
    c = _cases.IGEV_LOOP_CASES["kitti"]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    run = lambda cache: igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, 12, cache=cache)[0]
    ref = run({})
    monkeypatch.setattr(loop_c8, "SCHEDULE", (3, 4))
    cache = {}
    a = run(cache)
    b = run(cache)
    lp = cache["state"].c8
    assert torch.equal(a, b) and {k[0] for d in (lp.graph, lp.graph_n, lp.graph_last) for k in d} == {1, 2, 3}
    d = maxabs(a, ref)
    print("IGEV schedule (3, 4) vs fp32-class: %.3e" % d)
    assert 0 < d <= 5e-2
    monkeypatch.setattr(loop_c8, "SCHEDULE", None)
    assert torch.equal(run({}), ref)
