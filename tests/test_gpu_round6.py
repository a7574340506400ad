"""-m gpu: round 6.

* the post-conditions of a pair in one launch (csrc/status.hip, dkt_loop_status) against torch's reductions, and through the
  raw ABI; the checked forward on top of it: a GRU time-out (bit 0) and a chain time-out (bit 1) fall back separately, direct
  callers of encode() + iterate() are covered too;
* args.mixed_precision (raft_stereo.py:95,156; tools/ft_dkt.py:317): False stays the parity path bit for bit, True runs the
  one-product schedule, encoders included, with its measured distance from the reference fixture;
* IGEV's loop under nn.DataParallel replicas on a persistent per-device copy (igev_loop._ShadowBlock);
* the calibration stress test: 20 different pairs of very different magnitude through ONE model in
  default mode;
* bench.py's default-mode value, error word and --distinct-pairs on a small shape.
"""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest
import torch

import _cases
import _synth
from test_gpu_parity import DEV, G, _raft, maxabs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- dkt_loop_status ---------------------------------------------------------------------------------------------------------
@torch.no_grad()
def test_loop_status_matches_torch_reductions():
    """status() = the error word (cleared), finiteness of the result, and per tensor the maxima ActC8.absmax computes."""
    model, _ = _raft()
    i1, i2 = (G(t) for t in _synth.image_pair(4, 2, 96, 160, 12))
    _, up = model(i1, i2, iters=5, test_mode=True)
    lp = model._graph_state["c8"]
    s = lp.status(up)
    assert s.err == 0 and s.finite and s.ranges_ok
    for a, (body, tail) in zip(lp._range_acts(), s.maxima.tolist()):
        wb, wt = a.absmax()
        assert body == float(wb) and tail == (float(wt) if wt is not None else 0.0), (body, tail, float(wb), wt)
    assert 2.0 ** 6 <= s.maxima[:, 0].min() and s.maxima.max() < 2.0 ** 14
    # the error word comes back and is cleared by the same launch
    lp.err.fill_(3)
    s = lp.status(up)
    assert s.err == 3 and int(lp.err.item()) == 0 and lp.status(up).err == 0
    # a NaN / an Inf anywhere in the result; a non-contiguous view is checked through its base
    bad = up.clone()
    bad[1, 0, 5, 7] = float("inf")
    assert not lp.status(bad).finite
    full = torch.zeros(2, 2, 32, 32, device=DEV)
    full[1, 1, 3, 3] = float("nan")
    assert not lp.status(full[:, :1]).finite and lp.status(full[:1]).finite
    # an overflowed scale (Inf in a hi half) and a NaN are not mistaken for large numbers
    keep = lp.cor.t[0, 0, 0, 4, 4].clone()
    lp.cor.t[0, 0, 0, 4, 4, 1] = float("inf")
    s = lp.status(up)
    assert not s.ranges_ok and np.isinf(s.maxima[0, 0])
    lp.cor.t[0, 0, 0, 4, 4, 1] = float("nan")
    s = lp.status(up)
    assert not s.ranges_ok and np.isnan(s.maxima[0, 0])
    lp.cor.t[0, 0, 0, 4, 4] = keep
    # the tail channels (flow behind the 126 motion features) carry their own maximum
    t_before = float(lp.status(up).maxima[3, 1])
    g, k = 126 // 8, 126 % 8
    lp.mf.t[1, g, 0, 9, 9, k] = 3.0 * t_before
    s = lp.status(up)
    assert s.maxima[3, 1] == np.float32(np.float16(3.0 * t_before)) and s.maxima[3, 0] == float(lp.mf.absmax()[0])


def test_loop_status_argument_errors():
    from dkt_stereo_amd import _ffi, conv_c8
    L = _ffi.lib()
    a = conv_c8.ActC8(1, 64, 16, 32, DEV)
    st = torch.zeros(4, device=DEV, dtype=torch.int32)
    job = (_ffi.C8RangeJob * 1)()
    job[0].t, job[0].bstride_bytes, job[0].B, job[0].C, job[0].H, job[0].W, job[0].tail = a.data_ptr(), a.bstride_bytes, 1, 64, 16, 32, 0
    dev, stream = _ffi.device_of(st), _ffi.stream_of(st)
    assert L.dkt_loop_status(job, 1, None, 0, None, st.data_ptr(), dev, stream) == 0
    torch.cuda.synchronize()
    assert st.tolist() == [0, 0, 0, 0]
    assert L.dkt_loop_status(job, 1, None, 0, None, None, dev, stream) == -1                  # DKT_E_NULL
    assert L.dkt_loop_status(job, 1, None, 5, None, st.data_ptr(), dev, stream) == -1
    assert L.dkt_loop_status(job, _ffi.STATUS_MAX_JOBS + 1, None, 0, None, st.data_ptr(), dev, stream) == -2   # DKT_E_SHAPE
    job[0].tail = 65
    assert L.dkt_loop_status(job, 1, None, 0, None, st.data_ptr(), dev, stream) == -2
    job[0].tail = 0
    job[0].t = a.data_ptr() + 2
    assert L.dkt_loop_status(job, 1, None, 0, None, st.data_ptr(), dev, stream) == -6         # DKT_E_ALIGN


# ---- the checked forward ----------------------------------------------------------------------------------------------------
@torch.no_grad()
def test_chain_timeout_leaves_the_fused_gru_alone():
    """Bit 1 of the error word (a chain launch's time-out) switches the chains off, not the unrelated fused ConvGRU launch
    (ADVICE r05); bit 0 switches both off."""
    H, W = 544, 960
    i1, i2 = (G(t) for t in _synth.image_pair(6, 1, H, W, 12))
    model, _ = _raft()
    _, first = model(i1, i2, iters=4, test_mode=True)
    lp = model._graph_state["c8"]
    assert lp.fuse_gru and lp.chain_ok
    lp.err.fill_(2)
    with pytest.warns(UserWarning, match="chain launch timed out"):
        _, second = model(i1, i2, iters=4, test_mode=True)
    assert lp.fuse_gru and not lp.chain_ok and int(lp.err.item()) == 0
    assert torch.equal(second, first)                        # (chains are off by default: the same launches ran)
    lp.err.fill_(1)
    with pytest.warns(UserWarning, match="fused ConvGRU launch timed out"):
        _, third = model(i1, i2, iters=4, test_mode=True)
    assert not lp.fuse_gru and maxabs(third, first) <= 2e-4


@torch.no_grad()
def test_iterate_verifies_for_direct_callers():
    """encode() + iterate() called directly (bench.py's hot path) get the same post-condition check as forward():
    a raised error word is a loud _RetryForward (a DktError), after which the same calls give the fallback's result."""
    from dkt_stereo_amd import _ffi
    from dkt_stereo_amd.raft_stereo import _RetryForward
    H, W = 544, 960
    i1, i2 = (G(t) for t in _synth.image_pair(6, 1, H, W, 12))
    model, _ = _raft()
    _, want = model(i1, i2, iters=4, test_mode=True)
    lp = model._graph_state["c8"]
    lp.err.fill_(1)
    with pytest.warns(UserWarning, match="timed out"), pytest.raises(_RetryForward):
        model.iterate(*model.encode(i1, i2), 4)
    assert issubclass(_RetryForward, _ffi.DktError) and not lp.fuse_gru
    _, got = model.iterate(*model.encode(i1, i2), 4)
    assert maxabs(got, want) <= 2e-4
    # unchecked mode (what bench.py's timed region runs): no synchronisation, no exception; the word stays for the caller
    model.check_finite = False
    lp.err.fill_(1)
    model.iterate(*model.encode(i1, i2), 4)
    assert lp.status().err == 1


@torch.no_grad()
def test_fused_gru_one_product_with_an_odd_chunk_count_takes_two():
    """The one-product kernel walks the x chunks in pairs: a ConvGRU whose x operands give an odd number of 16-channel chunks gets
    the two-product form from gru_desc (as conv_c8.desc does for its layers) instead of a declined launch -- which used to switch
    the fused ConvGRU off for every later unit of the loop."""
    from test_gpu_round4 import _State, _make
    from dkt_stereo_amd import conv_c8 as c8
    gru, h, xs, cz, cr, cq = _make(1, 64, 96, [128, 16], seed=11)         # 8 + 1 chunks of x
    err = torch.zeros(1, device=DEV, dtype=torch.int32)
    a, b = _State(gru, h, xs, cz, cr, cq), _State(gru, h, xs, cz, cr, cq)
    with c8.passes(1):
        d1 = a.desc()
    with c8.passes(2):
        d2 = b.desc()
    assert d1.passes == 2 and d2.passes == 2
    assert c8.gru_launch(d1, err=err) and c8.gru_launch(d2, err=err)
    assert torch.equal(a.h, b.h) and int(err.item()) == 0
    gru2, h2, xs2, cz2, cr2, cq2 = _make(1, 64, 96, [128, 128], seed=12)    # an even count keeps the one-product form
    with c8.passes(1):
        assert _State(gru2, h2, xs2, cz2, cr2, cq2).desc().passes == 1


# ---- args.mixed_precision ------------------------------------------------------------------------------------------------------
@torch.no_grad()
def test_mixed_precision_flag(golden):
    """raft_stereo.py:95,156: `mixed_precision` wraps the encoders and the update block in fp16 autocast (default True on the DKT
    teachers, tools/ft_dkt.py:317).  Here True = one fp16 MFMA product per block in the encoders and in every unit of the loop;
    False (base.json) is untouched: the parity path bit for bit.  The distance of the True path from the reference's fp32
    output is measured and bounded (its own line in bench.py, never the headline)."""
    name = "256x512_it32"
    c = _cases.E2E_CASES[name]
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"]))
    base, _ = _raft()
    _, want = base(i1, i2, iters=c["iters"], test_mode=True)
    off, _ = _raft(mixed_precision=False)
    _, same = off(i1, i2, iters=c["iters"], test_mode=True)
    assert torch.equal(same, want) and off._graph_state["c8"].schedule is None
    on, _ = _raft(mixed_precision=True)
    _, got = on(i1, i2, iters=c["iters"], test_mode=True)
    lp = on._graph_state["c8"]
    assert lp.schedule == (c["iters"], 0) and set(lp.plan(c["iters"])) == {1}
    assert {k[0] for d in (lp.graph, lp.graph_n, lp.graph_last) for k in d} == {1}                   # every captured unit is the one-product kind
    g = golden("raft_e2e")
    s = int(g[name + "/stride"])
    d_on = maxabs(got[:, :, ::s, ::s], g[name + "/flow_up"])
    d_off = maxabs(want[:, :, ::s, ::s], g[name + "/flow_up"])
    print("mixed_precision=True: max|d| %.3e from the reference fixture (False: %.3e)" % (d_on, d_off))
    assert d_off <= 1e-3 and 1e-5 < d_on <= 1e-1
    _, again = on(i1, i2, iters=c["iters"], test_mode=True)
    assert torch.equal(again, got)
    # switching the key off on the same model restores the parity path
    on.args.mixed_precision = False
    _, back = on(i1, i2, iters=c["iters"], test_mode=True)
    assert maxabs(back, want) <= 2e-5 and on._graph_state["c8"].schedule is None


@torch.no_grad()
def test_mixed_precision_encoders_run_one_product():
    """Under mixed_precision the encoders' convolutions take the one-product backend (and only inside encode())."""
    from dkt_stereo_amd import conv
    i1, i2 = (G(t) for t in _synth.image_pair(3, 1, 64, 128, 12))
    on, _ = _raft(mixed_precision=True)
    off, _ = _raft()
    f_on = on.encode(i1, i2)[0]
    f_off = off.encode(i1, i2)[0]
    assert conv.get_backend() == "f16x3"
    rel = float((f_on - f_off).abs().max() / f_off.abs().max())
    assert 1e-5 < rel < 5e-2, rel                            # fp16-rounded operands: visible, and small


@torch.no_grad()
def test_mixed_precision_on_the_loops_the_c8_path_does_not_take():
    """slow_fast_gru / two GRU levels stay on the round-2 loop: with mixed_precision it runs on the one-product backend (`f16`), the
    key off it is the reference fixture's parity path, and the process-wide backend is untouched either way."""
    from dkt_stereo_amd import conv
    c = _cases.E2E_SLOWFAST_CASES["sf2_64x128_it6"]
    i1, i2 = (G(t) for t in _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"]))
    over = dict(slow_fast_gru=True, n_gru_layers=c["n"])
    off, _ = _raft(**over)
    on, _ = _raft(mixed_precision=True, **over)
    _, want = off(i1, i2, iters=c["iters"], test_mode=True)
    _, got = on(i1, i2, iters=c["iters"], test_mode=True)
    assert conv.get_backend() == "f16x3"
    st = on._graph_state
    assert st is None or st.get("c8") is None                 # (not the C8S loop)
    d = maxabs(got, want)
    print("mixed_precision on the round-2 loop: max|d| %.3e from the fp32-class path" % d)
    assert 1e-6 < d <= 1e-1
    _, again = on(i1, i2, iters=c["iters"], test_mode=True)
    assert torch.equal(again, got)


# ---- IGEV under nn.DataParallel replicas -----------------------------------------------------------------------------------------
@torch.no_grad()
def test_igev_data_parallel_replicas_run_on_a_persistent_shadow():
    """IGEV twin of test_data_parallel_replicas_run_on_a_persistent_shadow (tools/ft_dkt.py:119-125,193,199): replicas of the
    update block -- new modules on new threads for every forward -- hand igev_iterate to ONE persistent copy per device: the
    C8S loop is captured once and replayed, results equal the master's own call bit for bit, a weight update reaches the copy."""
    from test_gpu_round2 import _igev_setup
    from dkt_stereo_amd import igev_loop
    c = _cases.IGEV_LOOP_CASES["kitti"]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    iters = 6
    cache = {}
    want = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, iters, cache=cache)
    want = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, iters, cache=cache)
    assert cache["state"].c8 is not None

    def through_replicas(n):
        reps = [blk._replicate_for_data_parallel() for _ in range(n)]
        assert all(r._is_replica and r._dp_master() is blk for r in reps)
        out, err = [None] * n, []

        def run(k):
            try:
                out[k] = igev_loop.igev_iterate(reps[k], geo_fn, d0, coords, [t.clone() for t in net], inp, iters, cache={})
            except Exception as e:        # noqa: BLE001
                err.append(e)

        th = [threading.Thread(target=run, args=(k,)) for k in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        return out

    def same(a, b):
        return torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))

    for o in through_replicas(2):
        assert same(o, want)
    sh = igev_loop._SHADOWS[blk][torch.device(DEV).index]
    st = sh.cache["state"]
    lp = st.c8
    assert sh.block is not blk and lp is not None and lp.fuse_gru and lp.graph is not None
    graphs = dict(lp.graph)
    for o in through_replicas(2):
        assert same(o, want)
    assert sh.cache["state"] is st and sh.cache["state"].c8 is lp and dict(lp.graph) == graphs      # replayed, not re-captured
    # a weight update on the master (the EMA teacher changes every step) reaches the copy
    blk.disp_head.conv2.weight.mul_(1.5)
    want2 = igev_loop.igev_iterate(blk, geo_fn, d0, coords, [t.clone() for t in net], inp, iters, cache=cache)
    assert not torch.equal(want2[0], want[0])
    for o in through_replicas(2):
        assert same(o, want2)
    # the switch: replicas on their own run the plain loop
    igev_loop.REPLICA_SHADOWS = False
    try:
        o = through_replicas(1)[0]
    finally:
        igev_loop.REPLICA_SHADOWS = True
    assert maxabs(o[0], want2[0]) <= 1e-3


@torch.no_grad()
def test_igev_loop_rescales_when_the_disparity_leaves_the_window():
    """igev_iterate checks its call's post-conditions like RAFTStereo.iterate (round 6): an initial disparity 2^8 above the one the
    scales were picked on is rescaled from the maxima the call left behind and repeated -- the result equals a fresh state's."""
    from test_gpu_round2 import _igev_setup
    from dkt_stereo_amd import igev_loop
    c = _cases.IGEV_LOOP_CASES["kitti"]
    blk, geo_fn, d0, coords, net, inp, _ = _igev_setup(c)
    cache = {}
    run = lambda d, ca: igev_loop.igev_iterate(blk, geo_fn, d, coords, [t.clone() for t in net], inp, 6, cache=ca)
    run(d0, cache)
    lp = cache["state"].c8
    r0 = lp.recalibrations
    far = d0 * 600.0
    got = run(far, cache)
    assert lp.recalibrations == r0 + 1 and cache["state"].c8 is lp
    want = run(far, {})
    scale = max(1.0, float(want[0].abs().max()) / 256.0)
    assert maxabs(got[0], want[0]) <= 1e-3 * scale and maxabs(got[1], want[1]) <= 1e-3 * scale
    again = run(far, cache)
    assert lp.recalibrations == r0 + 1 and torch.equal(again[0], got[0])


@torch.no_grad()
def test_deferred_encoder_join_and_lazy_capture_change_nothing():
    """forward() leaves the join of the feature encoder's stream to the loop's prologue and captures each kind of unit on first
    use (round 6); encode() + iterate() called directly join in encode().  Same bits either way, and as a loop whose every kind of
    unit was captured up front."""
    i1, i2 = (G(t) for t in _synth.image_pair(8, 1, 544, 960, 40))
    a, _ = _raft()
    b, _ = _raft()
    b.defer_fnet_join = False
    for k in range(3):
        _, ua = a(i1, i2, iters=11, test_mode=True)
        _, ub = b(i1, i2, iters=11, test_mode=True)
        assert torch.equal(ua, ub), k
    _, uc = a.iterate(*a.encode(i1, i2), 11)
    assert torch.equal(uc, ua)
    lp = a._graph_state["c8"]
    kinds = {("one",) + k for k in lp.graph} | {("n",) + k for k in lp.graph_n} | {("last",) + k for k in lp.graph_last}
    assert 3 <= len(kinds) < 6, kinds                          # not every (kind, parity) is needed by an 11-iteration pair
    from dkt_stereo_amd.update import capture_graph
    lp.capture(a._graph_state, capture_graph, 3)
    assert len(lp.graph) + len(lp.graph_n) + len(lp.graph_last) == 6
    _, ud = a(i1, i2, iters=11, test_mode=True)
    assert torch.equal(ud, ua)


# ---- calibration stress -----------------------------------------------------------------------------------
@torch.no_grad()
def test_calibration_stress_twenty_pairs():
    """20 different pairs through ONE model in default mode -- image amplitudes 2^-3 ... 2^3 of the calibration pair's and
    disparities from 1 to 190 pixels in no order (the C8S scales follow the flow and the correlation features, not the image
    amplitude: it is the disparity range that leaves the window): every result within 1e-3 of the same pair on a fresh model
    (which picks its scales on that very pair), the number of recalibrations recorded, the worst forward bounded against the
    steady one."""
    H, W, iters = 256, 512, 12
    model, _ = _raft()
    base = [G(t) for t in _synth.image_pair(50, 1, H, W, 12)]
    for _ in range(3):
        model(base[0], base[1], iters=iters, test_mode=True)          # scales picked on the amplitude-1 pair; loop captured
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model(base[0], base[1], iters=iters, test_mode=True)
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 5
    lp = model._graph_state["c8"]
    r0, c0 = lp.recalibrations, lp.calibrations
    amps = [2.0 ** (-3 + 6 * k / 19) for k in range(20)]
    order = [0, 19, 1, 18, 10, 2, 17, 9, 3, 16, 11, 4, 15, 8, 5, 14, 12, 6, 13, 7]       # large jumps first
    shifts = [1, 190, 3, 120, 12, 40, 6, 80, 2, 160, 24, 60, 1, 190, 8, 100, 4, 140, 16, 48]
    times, dist, recal_at = [], [], []
    for j, k in enumerate(order):
        p1, p2 = _synth.image_pair(100 + k, 1, H, W, shifts[j])
        a1, a2 = G(p1) * amps[k], G(p2) * amps[k]
        before = lp.recalibrations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, got = model(a1, a2, iters=iters, test_mode=True)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if lp.recalibrations != before:
            recal_at.append(j)
        fresh, _ = _raft()
        _, want = fresh(a1, a2, iters=iters, test_mode=True)
        dist.append(maxabs(got, want))
        del fresh
    rec = lp.recalibrations - r0
    quiet = sorted(t for j, t in enumerate(times) if j not in recal_at)
    print("calibration stress: %d recalibrations over 20 pairs (at steps %s; %d without a trial run); steady %.2f ms, median of the "
          "forwards that kept their scales %.2f ms, worst %.2f ms (%.1fx steady); max distance to a fresh model %.2e"
          % (rec, recal_at, lp.calibrations - c0, 1e3 * steady, 1e3 * quiet[len(quiet) // 2], 1e3 * max(times), max(times) / steady,
             max(dist)))
    assert max(dist) <= 1e-3, dist
    assert model._graph_state["c8"] is lp                       # the same loop object served all of them
    assert rec <= 12                                            # (measured: 0 -- a 2^8 window around the calibration pair's maxima)
    assert max(times) <= 3.0 * steady, (max(times), steady)
    # A pair that does leave the window: an initial flow of -900 pixels (flow_init, raft_stereo.py:141-142) puts the flow
    # features 2^8 above what the scales were picked for.  The forward rescales from the maxima that pass left behind (no trial
    # run), re-captures and repeats the pair; going back to the first pair rescales again.
    want0 = model(base[0], base[1], iters=iters, test_mode=True)[1].clone()
    init = torch.zeros(1, 2, H // 4, W // 4, device=DEV)
    init[:, 0] = -900.0
    r1, c1 = lp.recalibrations, lp.calibrations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, got = model(base[0], base[1], iters=iters, flow_init=init, test_mode=True)
    torch.cuda.synchronize()
    t_recal = time.perf_counter() - t0
    assert lp.recalibrations == r1 + 1 and lp.calibrations == c1 + 1
    fresh, _ = _raft()
    _, want = fresh(base[0], base[1], iters=iters, flow_init=init, test_mode=True)
    d_far = maxabs(got, want)
    _, again = model(base[0], base[1], iters=iters, flow_init=init, test_mode=True)
    assert lp.recalibrations == r1 + 1 and torch.equal(again, got)            # the new scales hold for this kind of pair
    _, back = model(base[0], base[1], iters=iters, test_mode=True)
    assert lp.recalibrations == r1 + 2
    print("a pair 2^8 outside the window: recalibrated forward %.2f ms = %.1fx steady (rescale from the maxima, re-capture, the pair "
          "again); max distance to a fresh model %.2e; back on the first pair %.2e" % (1e3 * t_recal, t_recal / steady, d_far,
                                                                                      maxabs(back, want0)))
    assert d_far <= 1e-3 * max(1.0, float(want.abs().max()) / 256.0) and maxabs(back, want0) <= 1e-3
    # (encoders twice, the loop twice, and the captures of the unit kinds this pair replays; at this small shape a steady forward
    # is 6 ms and the captures dominate -- at the benchmark shape the same absolute cost is ~2.5 steady forwards)
    assert t_recal <= 15.0 * steady


# ---- bench.py ------------------------------------------------------------------------------------------------------------------
def test_bench_default_mode_error_word_and_distinct_pairs():
    """bench.py on a small shape: the line carries value_default_mode, error_word = 0, recalibrations and the distinct-pairs
    record."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--height", "128", "--width", "256", "--iters", "6", "--steps", "3",
                          "--warmup", "1", "--skip-cpu-baseline", "--distinct-pairs", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["error_word"] == 0 and j["value"] > 0 and j["value_default_mode"] > 0
    d = j["distinct_pairs"]
    assert d["pairs"] == 4 and len(d["step_ms"]) == 4 and d["recalibrations"] == j["recalibrations"] >= 0
    assert d["worst_step_ms"] >= d["median_step_ms"] > 0
