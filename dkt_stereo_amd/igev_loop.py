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
import torch

from .update import _side_stream


def _plain(update_block, geo_fn, disp, coords, net_list, inp_list, iters):
    n = update_block.args.n_gru_layers
    mask = None
    for itr in range(iters):
        geo_feat = geo_fn(disp, coords)
        net_list, mask, delta = update_block(net_list, inp_list, geo_feat, disp, iter16=(n == 3), iter08=(n >= 2),
                                             need_mask=(itr == iters - 1))
        disp = disp + delta
    return disp, mask, net_list


class _State:
    pass


def _body(ub, geo_fn, st, need_mask, last):
    """One iteration on static buffers.  Precondition (pipelined form): st.net[2] already holds this
    iteration's coarsest GRU update."""
    dev = st.disp.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    nets = list(st.net)
    ub.inplace_state = True
    saved = ub.side_stream
    ub.side_stream = False
    done_mid = torch.cuda.Event()
    try:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ub(nets, st.inp, iter04=False, iter08=True, iter16=False, update=False)        # mid GRU (i)
            done_mid.record(side)
            if not last:
                ub(nets, st.inp, iter04=False, iter08=False, iter16=True, update=False)    # coarse GRU (i+1)
        geo_feat = geo_fn(st.disp, st.coords)
        ub.before_fine = lambda: main.wait_event(done_mid)
        nets, mask, delta = ub(nets, st.inp, geo_feat, st.disp, iter16=False, iter08=False, need_mask=need_mask)
        main.wait_stream(side)
    finally:
        ub.before_fine = None
        ub.side_stream = saved
        ub.inplace_state = False
    st.disp.add_(delta)
    for dst, src in zip(st.net, nets):
        if dst is not src:
            dst.copy_(src)
    return mask


@torch.no_grad()
def igev_iterate(update_block, geo_fn, init_disp, coords, net_list, inp_list, iters, use_hip_graph=True, cache=None):
    """Returns (disp, mask_feat_4, net_list) after `iters` refinement iterations.
    `cache` (a dict the caller keeps, e.g. on its model) lets consecutive calls with the same shapes reuse
    the captured graph; without it every call captures anew."""
    n = update_block.args.n_gru_layers
    pipelined = (use_hip_graph and init_disp.is_cuda and iters >= 3 and n == 3
                 and not getattr(update_block.args, "slow_fast_gru", False) and update_block.side_stream)
    if not pipelined:
        return _plain(update_block, geo_fn, init_disp, coords, list(net_list), inp_list, iters)
    key = (init_disp.device, tuple(init_disp.shape), id(update_block), id(geo_fn))
    st = cache.get("state") if cache is not None else None
    if st is None or st.key != key:
        st = _State()
        st.key = key
        st.graph = None
        st.disp = init_disp.clone()
        st.coords = coords.clone()
        st.net = [t.clone() for t in net_list]
        st.inp = [[t.clone() for t in scale] for scale in inp_list]
        if cache is not None:
            cache["state"] = st
    else:
        st.disp.copy_(init_disp)
        st.coords.copy_(coords)
        for dst, src in zip(st.net, net_list):
            dst.copy_(src)
        for ds, ss in zip(st.inp, inp_list):
            for dst, src in zip(ds, ss):
                dst.copy_(src)
    ub = update_block
    ub.inplace_state = True
    try:                                             # prologue: coarsest GRU of iteration 0
        ub(list(st.net), st.inp, iter04=False, iter08=False, iter16=True, update=False)
    finally:
        ub.inplace_state = False
    done = 0
    if st.graph is None:
        _body(ub, geo_fn, st, False, False)          # eager once: packs weights, sizes the allocator
        done = 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _body(ub, geo_fn, st, False, False)
        st.graph = g                                 # (capturing records, it does not execute)
    for _ in range(iters - 1 - done):
        st.graph.replay()
    mask = _body(ub, geo_fn, st, True, True)
    return st.disp.clone(), mask, [t.clone() for t in st.net]
