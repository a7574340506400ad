"""The IGEV-Stereo refinement loop, ``meta_arch/igev_stereo/igev_stereo.py:192-210`` in test_mode,
from the point where this library's operators take over (geometry encoding volume, update block):

    geo_fn = Combined_Geo_Encoding_Volume(match_left, match_right, geo_encoding_volume, radius, num_levels)
    disp, mask_feat_4, net_list = igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters)

Same arithmetic and order of updates as the reference loop (``disp = disp + delta_disp`` after every
update-block call; the mask features of the last iteration are returned for ``upsample_disp``).
On a HIP device the iteration body is captured once per shape into a HIP graph and replayed, and the
two coarse GRUs are software-pipelined across iterations on a second stream exactly as in
``raft_stereo.RAFTStereo`` (gru08(i) beside geo lookup / motion encoder / gru04(i); gru16(i+1) right
after it) -- every GRU sees the inputs the sequential order gives it, results are bit-identical to the
plain loop.  The feature / 3-D aggregation networks that produce the inputs are not part of this
library.
"""
import os
import threading
import warnings
import weakref

import torch

from . import conv as _conv
from . import conv_c8 as c8
from . import loop_c8
from .conv import conv2d
from .update import FUSE_GATES, GPU_GUARD, _side_stream, capture_graph, gru_pair, harness, interp, pool2x, replay_graph

#: the coarsest GRU of the next iteration shares the finest GRU's two launches (dkt_conv2d_f16s_pair)
PAIR_GRUS = True


def _plain(update_block, geo_fn, disp, coords, net_list, inp_list, iters):
    """igev_stereo.py:199-210 as written, including the slow-fast schedule (:204-207): with
    ``slow_fast_gru`` the coarsest GRU (3 layers) and then the coarsest + middle GRUs are advanced
    before every full update."""
    n = update_block.args.n_gru_layers
    slow_fast = getattr(update_block.args, "slow_fast_gru", False)
    mask = None
    for itr in range(iters):
        geo_feat = geo_fn(disp, coords)
        if n == 3 and slow_fast:
            net_list = update_block(net_list, inp_list, iter16=True, iter08=False, iter04=False, update=False)
        if n >= 2 and slow_fast:
            net_list = update_block(net_list, inp_list, iter16=(n == 3), iter08=True, iter04=False, update=False)
        net_list, mask, delta = update_block(net_list, inp_list, geo_feat, disp, iter16=(n == 3), iter08=(n >= 2),
                                             need_mask=(itr == iters - 1))
        disp = disp + delta
    return disp, mask, net_list


class _State:
    pass


#: the loop on the round-3 convolution (loop_c8.C8LoopIGEV) when the update block and the image size allow it (DKT_C8=0: never)
USE_C8 = os.environ.get("DKT_C8", "1") != "0"


def _iterate_c8(ub, st, iters):
    """loop_c8.C8LoopIGEV on the state's static buffers: prologue (coarsest GRU of iteration 0), `iters` units -- the first
    eagerly, one captured, the rest replayed --, the up-sampling mask features from the final hidden state."""
    lp = getattr(st, "c8", None)
    d = dict(net=st.net, inp=st.inp, disp=st.disp, coords=st.coords, geo_fn=st.geo_fn)
    if lp is None:
        lp = st.c8 = loop_c8.C8LoopIGEV(ub, d)
    with harness(inplace_state=True, side_stream=False):
        if not lp.calibrated:
            lp.calibrate(d, iters)               # activation scales from a trial run on this pair (state restored)
        with c8.passes(lp.plan(iters)[0]):
            lp.prologue(d)
        # every unit also advances the coarsest GRU for the NEXT iteration (it rides in the finest GRU's launches); the caller
        # gets the state the reference's loop ends with: the coarsest state is saved in front of the last unit
        done = 0
        plan = lp.plan(iters)
        if lp.graph is None:
            if iters == 1:
                keep = st.net[2].clone()
            with c8.passes(plan[0]):
                lp.unit(d, last=(iters == 1))    # eager once: packs weights, sizes the allocator
            done = 1
        if iters > done:
            lp.replay(plan[done:-1], d, capture_graph)
            keep = st.net[2].clone()
            lp.replay(plan[-1:], d, capture_graph, last=True)
        st.net[2].copy_(keep)
        mask = conv2d(st.net[0], ub.mask_feat_4[0], relu=True)
    return st.disp.clone(), mask, [t.clone() for t in st.net]


def _body(ub, st, need_mask, last):
    """One iteration on static buffers.  Precondition (pipelined form): st.net[2] already holds this
    iteration's coarsest GRU update."""
    dev = st.disp.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    nets = list(st.net)
    done_mid = torch.cuda.Event()
    up_mid, pool_mid = [], []
    pair = PAIR_GRUS and not last
    with harness(inplace_state=True, side_stream=False, before_fine=lambda: main.wait_event(done_mid),
                 fine_interp=lambda: up_mid[0], pair_coarse=pair, coarse_pool=lambda: pool_mid[0],
                 accumulate_into=st.disp):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ub(nets, st.inp, iter04=False, iter08=True, iter16=False, update=False)        # mid GRU (i)
            up_mid.append(interp(nets[1], nets[0]))      # the finest GRU's up-sampled operand, off the critical path
            if pair:
                pool_mid.append(pool2x(nets[1]))         # the coarsest GRU (i+1) shares the finest GRU's launches
            if not torch.cuda.is_current_stream_capturing():
                for t in up_mid + pool_mid:
                    t.record_stream(main)
            done_mid.record(side)
            if not last and not pair:
                ub(nets, st.inp, iter04=False, iter08=False, iter16=True, update=False)    # coarse GRU (i+1)
        geo_feat = st.geo_fn(st.disp, st.coords)
        nets, mask, delta = ub(nets, st.inp, geo_feat, st.disp, iter16=False, iter08=False, need_mask=need_mask)
        main.wait_stream(side)
    assert delta is None                                 # the head added it to st.disp in its epilogue
    for dst, src in zip(st.net, nets):
        if dst is not src:
            dst.copy_(src)
    return mask


# ---- rotated schedule (as raft_stereo.RAFTStereo._iterate_rotated): the captured unit is
#   { middle GRU (i) on the side stream  ||  disparity head (i-1), geometry lookup (i), motion encoder (i) }  ->
#   finest GRU (i) + coarsest GRU (i+1) in shared launches
# -- the middle GRU of an iteration needs the finest state of the PREVIOUS one only.  Opt-in here (the module attribute ROTATE):
# measured no gain on the IGEV loop (32.4 ms either way at cfg3 -- its geometry lookup and 162-channel motion
# encoder already cover the middle GRU), so the default stays the schedule of _body.
ROTATE = False


def _mid(ub, nets, inp, hold):
    ub(nets, inp, iter04=False, iter08=True, iter16=False, update=False)
    hold["up"] = interp(nets[1], nets[0])
    hold["pool"] = pool2x(nets[1])


def _fine(ub, nets, inp, mf, hold):
    nets[0], nets[2] = gru_pair(ub.gru04, (nets[0], *inp[0], [mf, hold["up"]], nets[0]),
                                ub.gru16, (nets[2], *inp[2], [hold["pool"]], nets[2]))


def _head(ub, st, need_mask):
    ub.disp_head.add_to(st.net[0], st.disp)          # disp += head(net): the add rides in the tail layer's epilogue
    return conv2d(st.net[0], ub.mask_feat_4[0], relu=True) if need_mask else None


def _rotated_unit(ub, st):
    dev = st.disp.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    hold = {}
    done = torch.cuda.Event()
    with harness(inplace_state=True, side_stream=False):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _mid(ub, st.net, st.inp, hold)
            if not torch.cuda.is_current_stream_capturing():
                for t in hold.values():
                    t.record_stream(main)
            done.record(side)
        _head(ub, st, False)
        mf = ub.encoder(st.disp, st.geo_fn(st.disp, st.coords))
        main.wait_event(done)
        main.wait_stream(side)
        _fine(ub, st.net, st.inp, mf, hold)


def _iterate_rotated(ub, st, iters):
    hold = {}
    with harness(inplace_state=True, side_stream=False):
        ub(st.net, st.inp, iter04=False, iter08=False, iter16=True, update=False)      # coarsest GRU (0)
        _mid(ub, st.net, st.inp, hold)
        mf = ub.encoder(st.disp, st.geo_fn(st.disp, st.coords))
        _fine(ub, st.net, st.inp, mf, hold)
    done = 0
    if st.graph is None:
        _rotated_unit(ub, st)                        # eager once: packs weights, sizes the allocator
        done = 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            _rotated_unit(ub, st)
        st.graph = g
    for _ in range(iters - 1 - done):
        replay_graph(st.graph)
    with harness(inplace_state=True, side_stream=False):
        return _head(ub, st, True)


def _fingerprint(update_block):
    """Everything a captured iteration holds pointers to through a cached derivative (packed weight
    images, merged z|r weights, biases): a changed parameter or backend must force a new capture."""
    fp = [(_conv.get_backend(), FUSE_GATES)]
    for t in update_block.parameters():
        fp.append((t.data_ptr(), t._version))
    for m in update_block.modules():
        e = getattr(m, "dkt_in_exp", None)       # calibrated activation exponents are baked into the captured launches
        if e:
            fp.append(("in_exp", id(m), e))
    return tuple(fp)


#: a replica made by nn.DataParallel hands its loop to a persistent per-device copy of the master's update block
#: (_ShadowBlock): captured units and packed weights survive between forwards.  False: replicas run the plain loop
REPLICA_SHADOWS = True


class _ShadowBlock:
    """IGEV twin of raft_stereo._Shadow (round 6).  nn.DataParallel (tools/ft_dkt.py:119-125: the
    teachers run test_mode forwards under it, :193,199) hands every forward NEW replica modules on NEW threads, while the
    captured loop, the packed weight images and the side streams of this library are keyed by module, weight storage and
    thread.  A replica's igev_iterate is therefore run on ONE persistent copy of the master's update block per device, its
    weights refreshed in place when the master's have changed, driven by ONE worker thread that lives as long as the master:
    the C8S loop is captured once and replayed by every later forward, bit-identical to the master's own call."""

    def __init__(self, device):
        from concurrent.futures import ThreadPoolExecutor
        self.device = torch.device(device)
        self.block = None
        self.fingerprint = None
        self.cache = {}
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dkt-igev-replica-cuda%d" % self.device.index)

    @staticmethod
    def _tensors(m):
        return list(m.parameters()) + list(m.buffers())

    def _sync(self, master):
        src = self._tensors(master)
        fp = tuple((t.data_ptr(), t._version) for t in src)
        if self.block is None:
            grus = [m for m in (getattr(master, n, None) for n in ("gru16", "gru08", "gru04")) if m is not None]
            hidden = [int(g.convz.weight.shape[0]) for g in grus]
            self.block = type(master)(master.args, hidden_dims=hidden).to(self.device)
        if fp != self.fingerprint:
            with torch.no_grad():
                for d, t in zip(self._tensors(self.block), src):
                    d.copy_(t)
            self.fingerprint = fp
        for a, b in zip(master.modules(), self.block.modules()):
            b.training = a.training
        for k, v in master.__dict__.items():                         # switches set on the instance (side_stream, ...)
            if not k.startswith("_") and k not in ("training", "args") and isinstance(v, (bool, int, float, str, tuple, type(None))):
                setattr(self.block, k, v)

    def _run(self, master, backend, ready, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph):
        torch.cuda.set_device(self.device)
        mine = torch.cuda.current_stream(self.device)
        mine.wait_event(ready)                       # the operands as the caller's stream left them
        with _conv.use_backend(backend), torch.no_grad(), GPU_GUARD.shared():
            self._sync(master)
            out = _igev_iterate(self.block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph, self.cache)
        done = torch.cuda.Event()
        done.record(mine)
        return out, done

    def run(self, master, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph):
        with torch.cuda.device(self.device):
            caller = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(caller)
            out, done = self.pool.submit(self._run, master, _conv.get_backend(), ready, geo_fn, init_disp, coords, net_list, inp_list,
                                         iters, use_hip_graph).result()
            caller.wait_event(done)
            disp, mask, nets = out
            for t in [disp, mask] + list(nets):      # allocated on the worker's stream, consumed on the caller's
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(caller)
        return out


_SHADOWS = weakref.WeakKeyDictionary()             # master update block -> {device index: _ShadowBlock}
_SHADOW_LOCK = threading.Lock()


def _shadow_of(master, device):
    with _SHADOW_LOCK:
        per = _SHADOWS.setdefault(master, {})
        sh = per.get(device.index)
        if sh is None:
            sh = per[device.index] = _ShadowBlock(device)
        return sh


@torch.no_grad()
def igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph=True, cache=None):
    """Returns (disp, mask_feat_4, net_list) after `iters` refinement iterations.
    `cache` (a dict the caller keeps, e.g. on its model) lets consecutive calls with the same shapes and
    weights reuse the captured graph (a new geometry volume of the same shapes is copied into the cached
    one's buffers; to avoid that copy keep ONE volume and call its ``rebuild``); without `cache` every call
    captures anew.  ``slow_fast_gru`` runs the plain loop (igev_stereo.py:204-207).  An nn.DataParallel replica of an update
    block runs on the master's persistent per-device copy (_ShadowBlock), whatever `cache` it brings."""
    if (REPLICA_SHADOWS and use_hip_graph and getattr(update_block, "_is_replica", False) and init_disp.is_cuda
            and not _conv.calibrating()):
        master = getattr(update_block, "_dp_master", lambda: None)()
        if master is not None:
            return _shadow_of(master, init_disp.device).run(master, geo_fn, init_disp, coords, net_list, inp_list, iters,
                                                            use_hip_graph)
    with GPU_GUARD.shared():
        return _igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph, cache)


def _igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph, cache):
    n = update_block.args.n_gru_layers
    # (conv.calibrate() records activation ranges with host synchronisation: plain loop, every iteration observed)
    pipelined = (use_hip_graph and not _conv.calibrating() and not getattr(update_block, "_is_replica", False) and init_disp.is_cuda and iters >= 3 and n == 3
                 and not getattr(update_block.args, "slow_fast_gru", False) and update_block.side_stream)
    if not pipelined:
        return _plain(update_block, geo_fn, init_disp, coords, list(net_list), inp_list, iters)
    key = (init_disp.device, tuple(init_disp.shape), tuple(geo_fn._shape), geo_fn._w2, geo_fn.num_levels,
           geo_fn.radius, _fingerprint(update_block), ROTATE, PAIR_GRUS, USE_C8)
    st = cache.get("state") if cache is not None else None
    if st is None or st.key != key or st.ub() is not update_block:
        st = _State()
        st.key = key
        st.ub = weakref.ref(update_block)
        st.graph = None
        # the state keeps the geometry volume alive (the graph holds pointers into its pyramids); a new
        # volume of the same shapes -- the reference builds one per pair -- is copied into these buffers
        st.geo_fn = geo_fn
        if cache is not None:
            geo_fn.own_buffers()                     # before the capture: changes the level-0 pointer
        # the running disparity lives in the tail channel of the motion-feature buffer the encoder writes into
        # (no copy for the torch.cat of igev_stereo/update.py:92)
        b_, _, h_, w_ = init_disp.shape
        _, st.disp = update_block.encoder.new_feature_buffer(b_, h_, w_, init_disp.device)
        st.disp.copy_(init_disp)
        st.coords = coords.clone()
        st.net = [t.clone() for t in net_list]
        st.inp = [[t.clone() for t in scale] for scale in inp_list]
        if cache is not None:
            cache["state"] = st
    else:
        if geo_fn is not st.geo_fn:
            st.geo_fn.copy_from(geo_fn)
        st.disp.copy_(init_disp)
        st.coords.copy_(coords)
        for dst, src in zip(st.net, net_list):
            dst.copy_(src)
        for ds, ss in zip(st.inp, inp_list):
            for dst, src in zip(ds, ss):
                dst.copy_(src)
    ub = update_block
    if USE_C8 and loop_c8.eligible_igev(ub, st.net[0].shape):
        out = _iterate_c8(ub, st, iters)
        # the post-conditions of this call in one launch and one host read (csrc/status.hip, as RAFTStereo.iterate): the error
        # word of the flag-synchronised launches, finiteness of the disparity, the C8S scale window
        status = st.c8.status(out[0])
        word = status.err
        if not word and st.c8.calibrated and not (status.finite and status.ranges_ok) and getattr(st, "rescaled", 0) < 2:
            # this pair's activations left the window the scales were picked for (round 6: IGEV's loop never looked): new scales
            # -- from the maxima the call left behind when they are finite, a trial run otherwise -- and the call again
            import math
            st.rescaled = getattr(st, "rescaled", 0) + 1
            st.c8.recalibrations += 1
            if status.finite and all(math.isfinite(v) for v in status.maxima.reshape(-1).tolist()):
                st.c8.rescale_from(status.maxima)
                st.c8.calibrations += 1
            else:
                st.c8.calibrated = False
            try:
                return _igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph,
                                     cache if cache is not None else dict(state=st))
            finally:
                st.rescaled = 0
        if not word and not status.finite:
            raise _conv._ffi.DktError("igev_iterate produced a non-finite disparity: an activation left the range of the "
                                      "split-fp16 convolutions (or the inputs were not finite)")
        if word:
            # a fused ConvGRU (bit 0) or chain (bit 1) launch gave up waiting for a neighbour tile (csrc/gru_c8.hip; ADVICE
            # r04): the result is wrong.  That form is left for good, and this call is computed again from the caller's
            # (untouched) inputs.
            if getattr(st.c8, "timed_out", 0) & word:
                raise _conv._ffi.DktError("the IGEV loop reported a flag time-out in a form that had already been switched off")
            st.c8.timed_out = getattr(st.c8, "timed_out", 0) | word
            warnings.warn("dkt_stereo_amd: a %s launch timed out waiting for a neighbour tile; falling back to separate "
                          "launches for this update block (one fused-GRU model per device)" % ("fused ConvGRU" if word & 1 else "chain"))
            st.c8.on_error_word(word)
            return _igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph,
                                 cache if cache is not None else dict(state=st))
        return out
    if ROTATE and PAIR_GRUS:
        mask = _iterate_rotated(ub, st, iters)
        return st.disp.clone(), mask, [t.clone() for t in st.net]
    with harness(inplace_state=True):                # prologue: coarsest GRU of iteration 0
        ub(list(st.net), st.inp, iter04=False, iter08=False, iter16=True, update=False)
    done = 0
    if st.graph is None:
        _body(ub, st, False, False)                  # eager once: packs weights, sizes the allocator
        done = 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            _body(ub, st, False, False)
        st.graph = g                                 # (capturing records, it does not execute)
    for _ in range(iters - 1 - done):
        replay_graph(st.graph)
    mask = _body(ub, st, True, True)
    return st.disp.clone(), mask, [t.clone() for t in st.net]
