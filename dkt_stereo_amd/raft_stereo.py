"""End-to-end RAFT-Stereo inference harness around the HIP hot path.

Counterpart of ``meta_arch/raft_stereo/raft_stereo.py`` (``RAFTStereo.forward``
:85-187) with identical sub-module names -- ``cnet``, ``fnet``, ``update_block``,
``context_zqr_convs`` -- hence identical ``state_dict()`` keys.  The reference's
own ``RAFTStereo`` can equally be pointed at ``dkt_stereo_amd.corr`` /
``dkt_stereo_amd.update``; this class exists so that the whole
path can run, be timed and be parity-checked on a machine where the reference
is not present.

``forward(image1, image2, iters, flow_init, test_mode)`` follows the reference
call convention (tools/evaluate_stereo.py:129).  ``test_mode=True`` is the inference
hot path (captured loop, fused epilogues); ``test_mode=False`` with autograd enabled
is the training forward (tools/ft_dkt.py:223): every iteration's prediction, the
correlation block and the update operator as autograd nodes on this library's kernels.
"""
import itertools
import math
import os
import warnings
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ffi
from .conv import conv2d
from .corr import CORR_IMPLEMENTATIONS
from .extractor import BasicEncoder, MultiBasicEncoder, ResidualBlock
from .update import (FUSE_GATES, GPU_GUARD, BasicMultiUpdateBlock, _side_stream, capture_graph, gru_pair, harness, interp,
                     pool2x, replay_graph)
from . import conv as _conv
from . import extractor as _extractor
from .utils import coords_grid

#: configs/raft_stereo/base.json of the reference
BASE_CONFIG = dict(model="RAFTStereo", backbone_type="default", corr_implementation="reg",
                   shared_backbone=False, corr_levels=4, corr_radius=4, n_downsample=2,
                   context_norm="batch", slow_fast_gru=False, n_gru_layers=3,
                   hidden_dims=[128, 128, 128], mixed_precision=False)


def make_args(**overrides):
    cfg = dict(BASE_CONFIG)
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


class _Precomputed:
    """relu(convc1(lookup)) computed ahead of the motion encoder's call (stands in for the deferred lookup)."""

    def __init__(self, cor1):
        self.cor1 = cor1
        self.is_cuda = cor1.is_cuda
        self.device = cor1.device

    def materialize(self):
        raise RuntimeError("the lookup tensor was fused away (DKT_FUSE_LOOKUP=0 keeps it)")

    def conv1x1(self, layer, relu=True):
        return self.cor1


#: module -> {thread id: captured-iteration state}.  Kept outside the module (graphs and static buffers
#: are neither picklable nor deep-copyable) and per thread (two threads driving one module each need
#: their own static buffers and capture).
_GRAPH_STATES = weakref.WeakKeyDictionary()
_ENCODER_STATES = weakref.WeakKeyDictionary()      # same, for the captured encoder pass
_GRAPH_LOCK = threading.Lock()


_HANDOVER = threading.local()


class _RetryForward(_ffi.DktError):
    """Raised by RAFTStereo.iterate when the pair has to be computed again (the loop has switched form or scales)."""


_NONFINITE = ("RAFTStereo.forward produced non-finite disparities: an activation left the range of the "
              "split-fp16 convolutions (or the inputs were not finite).  Run one forward under "
              "dkt_stereo_amd.conv.calibrate() to set per-layer exponents, or use conv.set_backend('miopen').")


class _Shadow:
    """One persistent copy of a model on one device of an nn.DataParallel group, and the one thread that drives it.

    nn.DataParallel (tools/ft_dkt.py:119-125: the student, the teacher and its EMA twin; test_mode=True for the teachers,
    :193,199) hands every forward NEW replica modules with NEW parameter tensors, each on a NEW Python thread
    (torch.nn.parallel.parallel_apply).  The captured loop, the packed weight images and the per-thread side streams of
    this library are keyed by module, weight storage and thread: on a replica they would be rebuilt on every call, which is
    why replicas used to run the plain un-captured loop (missing #6).  Instead a replica's test_mode forward
    is handed to this object: a full copy of the master on the replica's device whose weights are refreshed (in place, so
    its caches notice) when the master's have changed, driven by a single worker thread that lives as long as the master,
    so that every thread-keyed state persists between forwards.  The worker issues on the device's current (default)
    stream, the one DataParallel's scatter leaves the inputs ready on."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.model = None
        self.fingerprint = None
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dkt-replica-cuda%d" % self.device.index)

    @staticmethod
    def _tensors(m):
        return list(itertools.chain(m.parameters(), m.buffers()))

    def _sync(self, master):
        src = self._tensors(master)
        fp = tuple((t.data_ptr(), t._version) for t in src)
        if self.model is None:
            self.model = type(master)(master.args).to(self.device)
            self.model._is_shadow = True
        if fp != self.fingerprint:
            with torch.no_grad():
                for d, t in zip(self._tensors(self.model), src):
                    d.copy_(t)
            self.fingerprint = fp
        for a, b in zip(master.modules(), self.model.modules()):
            b.training = a.training                                  # freeze_bn() and eval() of the master
        for k, v in master.__dict__.items():                         # switches set on the instance (use_hip_graph, ...)
            if not k.startswith("_") and k != "training" and isinstance(v, (bool, int, float, str, tuple, type(None))):
                setattr(self.model, k, v)                             # (tuple / None: precision_schedule)

    def _run(self, master, backend, ready, autocast, args, kwargs):
        torch.cuda.set_device(self.device)
        mine = torch.cuda.current_stream(self.device)
        mine.wait_event(ready)                       # the inputs as the caller's stream left them
        with _conv.use_backend(backend), torch.autocast("cuda", enabled=autocast[0], dtype=autocast[1]):
            self._sync(master)
            out = self.model(*args, **kwargs)
        done = torch.cuda.Event()
        done.record(mine)
        return out, done

    def run(self, master, *args, **kwargs):
        """The forward on the worker thread.  The worker issues on the device's default stream: it waits for an event the
        caller records on ITS current stream (parallel_apply's caller may be on another one), and the caller's stream waits
        for the worker's work before it touches the results; the caller's autocast state is carried over.  Memory: one
        extra copy of the weights plus the captured loop's buffers per device, also on the master's own device."""
        with torch.cuda.device(self.device):
            caller = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(caller)
            autocast = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"))
            out, done = self.pool.submit(self._run, master, _conv.get_backend(), ready, autocast, args, kwargs).result()
            caller.wait_event(done)
            # the results were allocated on the worker's (default) stream and are consumed on the caller's: the caching
            # allocator must not hand their blocks out again while the caller's stream still reads them
            for t in (out.values() if isinstance(out, dict) else out if isinstance(out, (tuple, list)) else [out]):
                for u in (t if isinstance(t, (tuple, list)) else [t]):
                    if torch.is_tensor(u) and u.is_cuda:
                        u.record_stream(caller)
        return out


_SHADOWS = weakref.WeakKeyDictionary()             # master module -> {device index: _Shadow}
_SHADOW_LOCK = threading.Lock()


def _shadow_of(master, device):
    with _SHADOW_LOCK:
        per = _SHADOWS.setdefault(master, {})
        sh = per.get(device.index)
        if sh is None:
            sh = per[device.index] = _Shadow(device)
        return sh


class RAFTStereo(nn.Module):
    @property
    def _graph_state(self):
        with _GRAPH_LOCK:
            return _GRAPH_STATES.get(self, {}).get(threading.get_ident())

    @_graph_state.setter
    def _graph_state(self, st):
        with _GRAPH_LOCK:
            per = _GRAPH_STATES.setdefault(self, {})
            if st is None:
                per.pop(threading.get_ident(), None)
            else:
                per[threading.get_ident()] = st

    def __init__(self, args=None):
        super().__init__()
        self.args = args = args if args is not None else make_args()
        if args.backbone_type not in ('default', 'interpolate'):
            raise ValueError("backbone_type: 'default' or 'interpolate' (raft_stereo.py:43-54)")
        context_dims = args.hidden_dims
        self.cnet = MultiBasicEncoder(output_dim=[args.hidden_dims, context_dims],
                                      norm_fn=args.context_norm, downsample=args.n_downsample)
        self.update_block = BasicMultiUpdateBlock(args, hidden_dims=args.hidden_dims)
        self.context_zqr_convs = nn.ModuleList(
            [nn.Conv2d(context_dims[i], args.hidden_dims[i] * 3, 3, padding=3 // 2)
             for i in range(args.n_gru_layers)])
        # raft_stereo.py:43-54: the default two-encoder backbone; `shared_backbone` = the context encoder's trunk on both
        # images + a residual block and a 3x3 layer (parameters conv2.0.*, conv2.1.*); 'interpolate' = no feature encoder,
        # the correlation runs on the bilinearly down-sampled images themselves
        if args.backbone_type == 'default':
            if args.shared_backbone:
                self.conv2 = nn.Sequential(ResidualBlock(128, 128, 'instance', stride=1), nn.Conv2d(128, 256, 3, padding=1))
            else:
                self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', downsample=args.n_downsample)

    @property
    def _two_encoders(self):
        return self.args.backbone_type == 'default' and not self.args.shared_backbone

    def _variant_features(self, image1, image2, n):
        """(cnet_list, fmap1, fmap2) of the non-default backbones, raft_stereo.py:97-108 (images already normalised)."""
        if self.args.backbone_type == 'default':                    # shared_backbone
            *cnet_list, x = self.cnet(torch.cat((image1, image2), dim=0), dual_inp=True, num_layers=n)
            f = self.conv2[1]
            y = self.conv2[0](x)
            y = conv2d(y, f) if (y.is_cuda and not torch.is_grad_enabled()) else f(y)
            fmap1, fmap2 = y.split(dim=0, split_size=x.shape[0] // 2)
        else:                                                       # 'interpolate'
            cnet_list = self.cnet(image1, num_layers=n)
            dw = 1 / (2 ** self.args.n_downsample)
            fmap1 = F.interpolate(image1, scale_factor=(dw, dw), mode='bilinear', align_corners=True)
            fmap2 = F.interpolate(image2, scale_factor=(dw, dw), mode='bilinear', align_corners=True)
        return list(cnet_list), fmap1.contiguous(), fmap2.contiguous()

    #: a replica made by nn.DataParallel runs its test_mode forward on a persistent per-device copy of the master (_Shadow):
    #: captured loop and packed weights survive between forwards.  False: replicas run the plain loop themselves
    replica_shadows = True

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica._dp_master = weakref.ref(self)
        return replica

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    #: run fnet and cnet on two HIP streams (DKT_ENCODER_STREAMS=0 disables)
    encoder_streams = os.environ.get("DKT_ENCODER_STREAMS", "1") != "0"

    #: forward(): the join of the feature encoder's stream is left to the loop's prologue (DKT_DEFER_FNET_JOIN=0: encode() joins)
    defer_fnet_join = os.environ.get("DKT_DEFER_FNET_JOIN", "1") != "0"

    #: enqueue the context encoder's full-resolution stage before the feature encoder (DKT_CNET_FIRST=0: after it, as rounds 2-5)
    cnet_first = os.environ.get("DKT_CNET_FIRST", "1") != "0"

    #: replay the encoder pass (normalisation, fnet || cnet, context split: ~110 launches, two streams) from a
    #: captured HIP graph as well: launched eagerly, the host needs 2.5 ms to enqueue fnet before cnet's first
    #: kernel can start, and ~0.3 ms of launch gaps precede the first convolution.  Opt-in (attribute graph_encoders): at
    #: 736x1248 the encoders are GPU-bound and the replay measured no faster (9.63 vs 9.45 ms); it pays on small images
    graph_encoders = os.environ.get("DKT_GRAPH_ENCODERS", "0") == "1"

    def _encoder_fingerprint(self):
        fp = [(_conv.get_backend(), _extractor.FUSE_ENCODER, self.encoder_streams)]
        for mod in ([self.fnet] if self._two_encoders else [m for m in [getattr(self, "conv2", None)] if m is not None]) + [self.cnet, self.context_zqr_convs]:
            for t in list(mod.parameters()) + list(mod.buffers()):
                fp.append((t.data_ptr(), t._version))
            for m in mod.modules():
                e = getattr(m, "dkt_in_exp", None)
                if e:
                    fp.append(("in_exp", id(m), e))
                if isinstance(m, nn.BatchNorm2d):
                    fp.append(("bn", id(m), m.training))
        return tuple(fp)

    @property
    def _mixed(self):
        """args.mixed_precision (raft_stereo.py:95,156: fp16 autocast around the encoders and the update block; default True on
        the DKT teachers, tools/ft_dkt.py:317).  Here: every convolution of the encoders and of the refinement loop at ONE fp16
        MFMA product per block (weights and activations rounded to fp16, fp32 accumulation and fp32 tensors between the
        layers -- at least the precision autocast keeps); correlation volume, lookup and up-sampling stay fp32, as the
        reference's `.float()` (raft_stereo.py:118-120) makes them.  False (base.json): the fp32-class parity path."""
        return bool(getattr(self.args, "mixed_precision", False))

    # -- pieces of the reference forward, split so the hot path can be timed alone --
    def encode(self, image1, image2):
        """raft_stereo.py:91-116: normalisation, encoders, context split.  The returned hidden states and context terms are
        valid until the next encode() or iterate() on this thread: once the captured loop exists for this shape they ARE
        its state buffers (adopt_encoder_outputs; with graph_encoders the captured pass's static outputs)."""
        if self._mixed and image1.is_cuda and _conv.get_backend() == "f16x3" and not torch.is_grad_enabled():
            with _conv.use_backend("f16"):
                return self._encode_any(image1, image2)
        return self._encode_any(image1, image2)

    def _encode_any(self, image1, image2):
        if (self.use_hip_graph and self.graph_encoders and not getattr(self, "_is_replica", False) and image1.is_cuda and image1.dtype == torch.float32
                and image2.dtype == torch.float32 and image1.shape == image2.shape
                and not torch.is_grad_enabled() and not _conv.calibrating()):
            return self._encode_graphed(image1, image2)
        return self._encode(image1, image2)

    def _encode_graphed(self, image1, image2):
        target = self._prebuild_target(image1)
        key = (image1.device, tuple(image1.shape), self.args.n_gru_layers, self._encoder_fingerprint(),
               None if target is None else (id(target), tuple(p.data_ptr() for p in target.corr_pyramid)))
        tid = threading.get_ident()
        with _GRAPH_LOCK:
            st = _ENCODER_STATES.setdefault(self, {}).get(tid)
        if st is None or st["key"] != key:
            out = self._encode(image1, image2)      # eager once: packs weights, folds norms, sizes the allocator
            torch.cuda.synchronize(image1.device)
            st = dict(key=key, img1=image1.clone(), img2=image2.clone())
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                st["out"] = self._encode(st["img1"], st["img2"])
            st["graph"] = g
            st["prebuilt"] = self._prebuilt          # the correlation block the captured pass rebuilds (or None)
            with _GRAPH_LOCK:
                _ENCODER_STATES.setdefault(self, {})[tid] = st
            self._static_ctx = None                  # (this call's tensors are the eager pass's, not the graph's)
            return out
        st["img1"].copy_(image1)
        st["img2"].copy_(image2)
        replay_graph(st["graph"])
        self._prebuilt = st["prebuilt"]
        self._static_ctx = st["out"]                 # the graph's static outputs: iterate() may adopt them as its state buffers
        return st["out"]

    #: the all-pairs volume + pyramid + skewed copy of the pair are built at the end of the feature encoder's stream, inside
    #: the captured encoder pass and beside the context encoder's tail of small launches, instead of behind both encoders
    prebuild_corr = True
    #: the refinement loop works in the captured encoder pass's own output tensors (hidden states, context terms) instead of
    #: copying them into buffers of its own (13 device copies per pair)
    adopt_encoder_outputs = True
    # hand-over from encode() to iterate() on the same thread (per thread, like the captured states themselves)
    def _handover(name):
        def get(self):
            return getattr(_HANDOVER, "d", {}).get((id(self), name))

        def put(self, value):
            d = _HANDOVER.__dict__.setdefault("d", {})
            if value is None:
                d.pop((id(self), name), None)
            else:
                d[(id(self), name)] = value
        return property(get, put)

    _prebuilt = _handover("prebuilt")        # the correlation block encode() has rebuilt for the pair at hand
    _defer_join = _handover("defer")         # forward() lets encode() leave the feature encoder's stream un-joined ...
    _pending_join = _handover("join")        # ... and this is that stream, for iterate() to join
    _static_ctx = _handover("static")        # encode()'s outputs when they are the captured pass's static tensors
    del _handover

    def _prebuild_target(self, image1):
        """The persistent correlation block of the captured loop, when this pair's volume can be rebuilt into it."""
        st = self._graph_state
        if not (self.prebuild_corr and self.use_hip_graph and st is not None and image1.is_cuda):
            return None
        corr = st.get("corr")
        f = st.get("fmap_shape")
        n = 2 ** self.args.n_downsample
        if corr is None or f is None or type(corr).__name__ != "CorrBlock1D":
            return None
        if (f[0], f[2] * n, f[3] * n) != (image1.shape[0], image1.shape[2], image1.shape[3]):
            return None
        return corr

    @staticmethod
    def _normalized_pair(image1, image2):
        """(image1', image2', both) with x' = 2 * (x / 255) - 1 (raft_stereo.py:91-92) and both = cat([image1', image2']) --
        one kernel (dkt_normalize_pair) writing the concatenated batch the feature encoder consumes; the halves are views."""
        if (image1.is_cuda and image1.dtype == torch.float32 and image2.dtype == torch.float32 and image1.shape == image2.shape
                and image1.dim() == 4 and not torch.is_grad_enabled()):
            B = image1.shape[0]
            a = image1 if image1[0].is_contiguous() else image1.contiguous()
            b = image2 if image2[0].is_contiguous() else image2.contiguous()
            both = torch.empty((2 * B,) + tuple(image1.shape[1:]), device=image1.device, dtype=torch.float32)
            rc = _ffi.lib().dkt_normalize_pair(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), both.data_ptr(), B,
                                               a[0].numel(), _ffi.device_of(a), _ffi.stream_of(a))
            _ffi.check(rc, "dkt_normalize_pair")
            return both[:B], both[B:], both
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        return image1, image2, None

    def _encode(self, image1, image2):
        image1, image2, both = self._normalized_pair(image1, image2)
        n = self.args.n_gru_layers
        if not self._two_encoders:
            self._prebuilt = None
            cnet_list, fmap1, fmap2 = self._variant_features(image1, image2, n)
            cnet_list = [self._context_post(i, list(outs)) for i, outs in enumerate(cnet_list)]
            return fmap1.float(), fmap2.float(), [x[0] for x in cnet_list], [x[1] for x in cnet_list]
        fnet_in = [image1, image2] if both is None else both
        split = lambda f: f if both is None else f.split(split_size=image1.shape[0], dim=0)
        if self.encoder_streams and image1.is_cuda:
            # the two encoders are independent: fnet runs on a second stream beside cnet (the
            # small layers of either one leave CUs idle); joined before anything consumes fmaps
            main = torch.cuda.current_stream(image1.device)
            side = _side_stream(image1.device)
            side.wait_stream(main)
            # The host needs ~2 ms to enqueue the feature encoder's ~45 launches.  When the previous pair has been waited for
            # (check_finite: the default mode) the device would run the feature encoder's chain ALONE for that long -- its
            # statistics reductions and finalize launches leave most CUs idle -- so the context encoder's full-resolution
            # stage (5 launches, ~1.1 ms of work) goes on the main queue first (round 6: value_default_mode)
            begun = self.cnet._trunk_begin(image1) if self.cnet_first else None
            with torch.cuda.stream(side):
                fmap1, fmap2 = split(self.fnet(fnet_in))
                self._prebuild(image1, fmap1, fmap2)
            cnet_list = self.cnet(image1, num_layers=n, head_post=self._context_post, begun=begun)
            if (self._defer_join and self._prebuilt is not None and self.defer_fnet_join
                    and not torch.cuda.is_current_stream_capturing()):          # (a capture must end with its streams joined)
                # forward(): the feature maps' only consumer -- the correlation build -- has already been enqueued on the
                # feature encoder's stream; the loop's prologue forks ITS correlation-dependent half (lookup + motion encoder of
                # iteration 0) onto that same stream and joins it itself, so the hidden-state half of the prologue (pack, gru32,
                # gru16: ~0.4 ms) starts as soon as the context encoder is done instead of behind the feature encoder's tail
                # (conv2, correlation build, skewed copy: ~0.35 ms).  iterate() joins wherever that does not hold.
                self._pending_join = side
            else:
                main.wait_stream(side)
        else:
            self._prebuilt = None
            cnet_list = self.cnet(image1, num_layers=n, head_post=self._context_post)
            fmap1, fmap2 = split(self.fnet(fnet_in))
        net_list = [x[0] for x in cnet_list]
        inp_list = [x[1] for x in cnet_list]
        return fmap1.float(), fmap2.float(), net_list, inp_list

    def _prebuild(self, image1, fmap1, fmap2):
        """core/corr.py:148-156 + :111-125 for this pair into the captured loop's correlation block, on the feature
        encoder's stream (see prebuild_corr); iterate() finds it done."""
        self._prebuilt = None
        if torch.is_grad_enabled():
            return               # (rebuild() under autograd would go through _BuildFn and replace the captured loop's pyramid tensors)
        corr = self._prebuild_target(image1)
        if corr is not None and tuple(fmap1.shape) == tuple(self._graph_state["fmap_shape"]):
            f1, f2 = fmap1.float(), fmap2.float()
            corr.rebuild(f1, f2)
            # the hand-over names the feature maps it was built from: iterate() skips its own rebuild only for THESE (ADVICE r04:
            # encode(A); encode(B); iterate(fmaps_A) must not run on B's volume)
            self._prebuilt = (corr, f1.data_ptr(), f2.data_ptr())

    def _context_post(self, i, outs):
        """raft_stereo.py:103-106 for scale i: tanh of the hidden head, relu + context_zqr convolution of the context
        head (split into the cz, cr, cq operands of that scale's GRU) -- run by the context encoder on the stream of
        the scale's heads."""
        conv = self.context_zqr_convs[i]
        tgt = self._context_targets(i, outs)
        if tgt is not None:
            # straight into the captured loop's state buffers (hidden state; the three context terms as views of one
            # buffer): iterate() finds them in place and skips its 12 device copies per pair
            net, ctx = tgt
            torch.tanh(outs[0], out=net)
            conv2d(torch.relu(outs[1]), conv, out=ctx)
            return [net, list(ctx.split(split_size=conv.out_channels // 3, dim=1))]
        return [torch.tanh(outs[0]),
                list(conv2d(torch.relu(outs[1]), conv).split(split_size=conv.out_channels // 3, dim=1))]

    def _context_targets(self, i, outs):
        """(hidden-state buffer, context buffer) of scale i in the captured loop's state, when this pass may write them
        (same thread's state, same shapes, inference on the device)."""
        st = self._graph_state
        if not (self.adopt_encoder_outputs and self.use_hip_graph and st is not None and outs[0].is_cuda
                and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            return None
        ctx = st.get("ctx")
        if ctx is None or i >= len(ctx):
            return None
        net, buf = st["net"][i], ctx[i]
        conv = self.context_zqr_convs[i]
        if (net.shape != outs[0].shape or net.device != outs[0].device or buf.shape[1] != conv.out_channels
                or buf.shape[0] != outs[1].shape[0] or buf.shape[2:] != outs[1].shape[2:] or outs[0].dtype != torch.float32):
            return None
        return net, buf

    def upsample_flow(self, flow, mask):
        """raft_stereo.py:70-82, convex combination over a 3x3 neighbourhood: one fused kernel
        (dkt_convex_upsample) on the GPU inference path, the reference's op sequence otherwise."""
        N, D, H, W = flow.shape
        factor = 2 ** self.args.n_downsample
        if flow.is_cuda and not (torch.is_grad_enabled() and (flow.requires_grad or mask.requires_grad)):
            flow = flow.float().contiguous()
            mask = mask.float().contiguous()
            out = torch.empty((N, D, factor * H, factor * W), device=flow.device, dtype=torch.float32)
            rc = _ffi.lib().dkt_convex_upsample(flow.data_ptr(), mask.data_ptr(), out.data_ptr(), N, D, H, W,
                                                factor, _ffi.device_of(flow), _ffi.stream_of(flow))
            _ffi.check(rc, "dkt_convex_upsample")
            return out
        mask = torch.softmax(mask.view(N, 1, 9, factor, factor, H, W), dim=2)
        up = F.unfold(factor * flow, [3, 3], padding=1).view(N, D, 9, 1, 1, H, W)
        up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
        return up.reshape(N, D, factor * H, factor * W)

    #: replay the GRU iteration from a captured HIP graph (one capture per input shape and weight set;
    #: DKT_HIP_GRAPH=0 runs every iteration eagerly)
    use_hip_graph = os.environ.get("DKT_HIP_GRAPH", "1") != "0"
    #: hand the motion encoder a deferred lookup: lookup + convc1 run as one kernel
    #: (dkt_corr1d_lookup_conv1x1); DKT_FUSE_LOOKUP=0 keeps the two launches
    fuse_lookup = os.environ.get("DKT_FUSE_LOOKUP", "1") != "0"

    def _lookup(self, corr_fn, coords1):
        if self.fuse_lookup and hasattr(corr_fn, "deferred"):
            return corr_fn.deferred(coords1)
        return corr_fn(coords1)

    def _one_iteration(self, corr_fn, coords0, coords1, net_state, inp_list, need_mask):
        """One pass of raft_stereo.py:146-167 updating coords1 / net_state IN PLACE
        (static buffers, so the same code can be captured once and replayed)."""
        args = self.args
        n = args.n_gru_layers
        corr = self._lookup(corr_fn, coords1)
        flow = coords1 - coords0
        nets = list(net_state)
        with harness(inplace_state=True):
            if n == 3 and args.slow_fast_gru:
                nets = self.update_block(nets, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if n >= 2 and args.slow_fast_gru:
                nets = self.update_block(nets, inp_list, iter32=(n == 3), iter16=True, iter08=False, update=False)
        with harness(inplace_state=True):          # net_state tensors are private to this loop
            nets, up_mask, delta_flow = self.update_block(nets, inp_list, corr, flow, iter32=(n == 3),
                                                          iter16=(n >= 2), need_mask=need_mask)
        # stereo: project onto the epipolar line (raft_stereo.py:165-166: delta_flow[:,1] = 0; coords1 += delta)
        # -- one launch: only the x plane changes (y + 0.0 is y)
        coords1[:, :1].add_(delta_flow[:, :1])
        for dst, src in zip(net_state, nets):
            if dst is not src:
                dst.copy_(src)
        return up_mask

    #: software-pipeline the GRU stack across iterations (attribute pipeline_grus): gru16 of
    #: this iteration and gru32 of the NEXT one run on a second stream beside lookup + motion
    #: encoder + gru08 + flow head.  Every GRU still sees exactly the inputs the reference's
    #: sequential order gives it (gru32(i+1) needs only net[2](i), net[1](i)): bit-identical.
    pipeline_grus = True
    #: with the pipelined schedule: the motion encoder's flow branch on a third stream.  Opt-in
    #: (attribute branch_streams): measured gain 0.3 ms per pair, not worth a third capture branch by default
    branch_streams = False

    #: with the pipelined schedule: gru32 of the next iteration shares the two launches of gru08
    #: (dkt_conv2d_f16s_pair; pair_grus = False keeps it as launches of its own on the side stream)
    pair_grus = True

    def _can_pipeline(self):
        a = self.args
        return (self.pipeline_grus and a.n_gru_layers == 3 and not a.slow_fast_gru
                and self.update_block.side_stream)

    def _weights_fingerprint(self):
        """(data_ptr, version) of every tensor the captured iteration reads through a cached derivative."""
        fp = [(_conv.get_backend(), FUSE_GATES)]
        for t in self.update_block.parameters():
            fp.append((t.data_ptr(), t._version))
        for m in self.update_block.modules():
            e = getattr(m, "dkt_in_exp", None)
            if e:
                fp.append(("in_exp", id(m), e))
        return tuple(fp)

    def _one_iteration_pipelined(self, corr_fn, coords0, coords1, net_state, inp_list, need_mask, last):
        """raft_stereo.py:146-167 with the two coarse GRUs off the critical path.  Precondition:
        net_state[2] already holds gru32 of THIS iteration (prologue / previous call); unless
        `last`, gru32 of the next iteration is computed here.  Updates coords1 / net_state in place."""
        ub = self.update_block
        dev = coords1.device
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        nets = list(net_state)
        done16 = torch.cuda.Event()
        # the fork / join is done here (side_stream=False); the encoder runs on `main`, so its own
        # fork (branch_streams) is not nested inside another forked stream
        up16, pool16 = [], []
        pair = self.pair_grus and not last
        with harness(inplace_state=True, side_stream=False, branch_streams=self.branch_streams,
                     before_fine=lambda: main.wait_event(done16), fine_interp=lambda: up16[0],
                     pair_coarse=pair, coarse_pool=lambda: pool16[0]):
            # The lookup (fused with convc1) is launched BEFORE the fork: it is a 15-us latency-bound kernel, and
            # beside gru16's convolutions -- which occupy every CU's register file -- it would be stretched to
            # 50+ us without the pair finishing any earlier.
            corr = self._lookup(corr_fn, coords1)
            if hasattr(corr, "materialize"):
                cor1 = ub.encoder._cor1(corr)
                corr = _Precomputed(cor1)
            flow = coords1 - coords0
            side.wait_stream(main)                   # fork
            with torch.cuda.stream(side):
                ub(nets, inp_list, iter32=False, iter16=True, iter08=False, update=False)      # gru16(i)
                # gru08's up-sampled operand (core/update.py:127) is ready as soon as gru16 is: computed
                # here, off the critical path (it cost 32 us per iteration in front of gru08)
                up16.append(interp(nets[1], nets[0]))
                if pair:
                    pool16.append(pool2x(nets[1]))   # gru32(i+1)'s operand (core/update.py:119)
                if not torch.cuda.is_current_stream_capturing():
                    for t in up16 + pool16:
                        t.record_stream(main)
                done16.record(side)
                if not last and not pair:
                    ub(nets, inp_list, iter32=True, iter16=False, iter08=False, update=False)  # gru32(i+1)
            nets, up_mask, delta_flow = ub(nets, inp_list, corr, flow, iter32=False, iter16=False,
                                           need_mask=need_mask)
            main.wait_stream(side)                   # join
        # stereo: project onto the epipolar line (raft_stereo.py:165-166: delta_flow[:,1] = 0; coords1 += delta)
        # -- one launch: only the x plane changes (y + 0.0 is y)
        coords1[:, :1].add_(delta_flow[:, :1])
        for dst, src in zip(net_state, nets):
            if dst is not src:
                dst.copy_(src)
        return up_mask

    #: rotated software pipeline (rotate = False: the schedule of _one_iteration_pipelined).  The captured unit is
    #: not "iteration i" but  { gru16(i) on the side stream  ||  flow head(i-1), lookup(i), motion encoder(i) } ->
    #: gru08(i) + gru32(i+1) : the middle GRU of an iteration only needs the finest state of the PREVIOUS one, so
    #: it runs beside the previous iteration's flow head instead of in front of this iteration's finest GRU
    #: (trace: gru08 waited ~100 us per iteration for the gru16 chain).  Same operations on the same operands in
    #: an order the reference's data dependencies allow: bit-identical.
    rotate = True

    def _stage_mid(self, nets, inp, hold):
        """gru16 of the coming iteration + the two resampled copies of its result the other GRUs consume."""
        self.update_block(nets, inp, iter32=False, iter16=True, iter08=False, update=False)
        hold["up16"] = interp(nets[1], nets[0])          # gru08's operand (core/update.py:127)
        hold["pool16"] = pool2x(nets[1])                 # gru32's operand (core/update.py:119)

    def _stage_motion(self, corr_fn, coords0, coords1, flow=None):
        enc = self.update_block.encoder
        corr = self._lookup(corr_fn, coords1)
        if hasattr(corr, "materialize"):
            corr = _Precomputed(enc._cor1(corr))
        if flow is None:
            # the flow is computed straight into the tail channels of the motion-feature buffer (no copy for the
            # torch.cat of core/update.py:85)
            b, _, h, w = coords1.shape
            _, flow = enc.new_feature_buffer(b, h, w, coords1.device)
            torch.sub(coords1, coords0, out=flow)
        # else: `flow` is the tail of the loop's persistent feature buffer, kept equal to coords1 - coords0 by the
        # head's epilogue (_stage_head)
        return enc(flow, corr)

    def _stage_fine(self, nets, inp, mf, hold):
        """gru08 of this iteration and gru32 of the next one in shared launches (update.gru_pair)."""
        ub = self.update_block
        nets[0], nets[2] = gru_pair(ub.gru08, (nets[0], *inp[0], [mf, hold["up16"]], nets[0]),
                                    ub.gru32, (nets[2], *inp[2], [hold["pool16"]], nets[2]))

    #: the rotated loop computes only the x output of flow_head.conv2 (head_x_only = False: both, y dropped afterwards)
    head_x_only = True

    def _stage_head(self, nets, coords1, need_mask, coords0=None, flow=None):
        ub = self.update_block
        if self.head_x_only:
            # stereo: only x moves (raft_stereo.py:165-168) -- the y output is not computed, the x output is added to
            # coords1 in the tail layer's epilogue, and the same epilogue refreshes flow_x = coords1_x - coords0_x in the
            # persistent feature buffer (the y plane never changes)
            diff = None if flow is None else (coords0[:, :1], flow[:, :1])
            ub.flow_head.add_to(nets[0], coords1[:, :1], outputs=1, diff=diff)
        else:
            coords1[:, :1].add_(ub.flow_head(nets[0])[:, :1])
            if flow is not None:
                torch.sub(coords1, coords0, out=flow)
        mask = None
        if need_mask:
            mask = .25 * conv2d(conv2d(nets[0], ub.mask[0], relu=True), ub.mask[2])
        return mask

    def _rotated_unit(self, st):
        """flow head(i-1), lookup + motion encoder(i)  ||  gru16(i)   ->   gru08(i) + gru32(i+1)."""
        dev = st["coords1"].device
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        nets, hold = st["net"], {}
        done = torch.cuda.Event()
        with harness(inplace_state=True, side_stream=False):
            side.wait_stream(main)                       # fork
            with torch.cuda.stream(side):
                self._stage_mid(nets, st["inp"], hold)
                if not torch.cuda.is_current_stream_capturing():
                    for t in hold.values():
                        t.record_stream(main)
                done.record(side)
            self._stage_head(nets, st["coords1"], False, st["coords0"], st["flow"])
            mf = self._stage_motion(st["corr"], st["coords0"], st["coords1"], st["flow"])
            main.wait_event(done)                        # join
            main.wait_stream(side)
            self._stage_fine(nets, st["inp"], mf, hold)

    def _iterate_rotated(self, st, iters):
        """Prologue: gru32(0), gru16(0), lookup + motion encoder(0), gru08(0) + gru32(1).  Then iters-1 rotated
        units (the first eagerly, one captured, the rest replayed), then the last flow head with the mask."""
        nets, hold = st["net"], {}
        if "flow" not in st:                 # persistent motion-feature buffer; its tail holds flow = coords1 - coords0
            b, _, h, w = st["coords1"].shape
            st["feat"], st["flow"] = self.update_block.encoder.new_feature_buffer(b, h, w, st["coords1"].device)
        torch.sub(st["coords1"], st["coords0"], out=st["flow"])          # both planes once per pair (covers flow_init)
        with harness(inplace_state=True, side_stream=False):
            self.update_block(nets, st["inp"], iter32=True, iter16=False, iter08=False, update=False)
            self._stage_mid(nets, st["inp"], hold)
            mf = self._stage_motion(st["corr"], st["coords0"], st["coords1"], st["flow"])
            self._stage_fine(nets, st["inp"], mf, hold)
        done = 0
        if st["graph"] is None:
            self._rotated_unit(st)           # eager once: packs weights, sizes the allocator
            done = 1
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                self._rotated_unit(st)
            st["graph"] = g
        for _ in range(iters - 1 - done):
            replay_graph(st["graph"])
        with harness(inplace_state=True, side_stream=False):
            return self._stage_head(nets, st["coords1"], True, st["coords0"], st["flow"])

    #: round 3: the loop on the C8S convolution (loop_c8.py); DKT_C8=0 keeps the round-2 kernels and schedule
    use_c8 = os.environ.get("DKT_C8", "1") != "0"
    #: round 5: (k1, k2) = the first k1 refinement iterations at one fp16 MFMA product per block, the next k2 at two, the rest at
    #: the fp32-class three (loop_c8.SCHEDULE); None = fp32-class throughout (the default and the parity path); False = whatever
    #: DKT_C8_SCHEDULE says.  The reference's switch of this kind: `mixed_precision`, raft_stereo.py:95,156
    precision_schedule = False
    c8_eager = False

    def _iterate_c8(self, st, iters, pending=None):
        """loop_c8.C8Loop: prologue, `iters` units (the first eagerly, one captured, the rest replayed), mask head.
        `pending`: the feature encoder's stream when encode() has left it un-joined (the correlation volume is being built on it)."""
        from . import loop_c8
        lp = st.get("c8")
        if lp is None:
            lp = st["c8"] = loop_c8.C8Loop(self, st)
        st["c8_ran"] = True
        if "flow" not in st:
            b, _, h, w = st["coords1"].shape
            st["feat"], st["flow"] = self.update_block.encoder.new_feature_buffer(b, h, w, st["coords1"].device)
        torch.sub(st["coords1"], st["coords0"], out=st["flow"])
        ub = self.update_block
        from . import conv_c8
        if self.precision_schedule is not False:          # (False: whatever DKT_C8_SCHEDULE says, else fp32-class)
            lp.schedule = self.precision_schedule
        elif self._mixed:
            lp.schedule = (iters, 0)                      # args.mixed_precision: one fp16 product per block throughout
        else:
            lp.schedule = loop_c8.SCHEDULE
        plan = lp.plan(iters)                # MFMA passes per unit: 3 throughout unless a precision schedule is set
        # the prologue forks its correlation-dependent half onto _side_stream(slot 0) -- the very stream the volume is being built
        # on -- and joins it: no other join is needed then.  Everything else (a trial run, a one-stream prologue) joins first
        if pending is not None and not (lp.calibrated and loop_c8.FORK and loop_c8.PROLOGUE_FORK
                                        and pending is _side_stream(st["coords1"].device, slot=0)):
            self._join_encoder(pending)
        with harness(inplace_state=True, side_stream=False):
            if not lp.calibrated:
                lp.calibrate(st, iters)      # activation scales from a trial run on this pair (state restored)
            with conv_c8.passes(plan[0]):
                lp.prologue(st)
            done = 0
            if self.c8_eager:                # (bench.py's instrumented pass: every unit as plain launches)
                for k in range(iters):
                    with conv_c8.passes(plan[k]):
                        lp.unit(st, last=(k + 1 == iters))
                done = iters
            elif lp.graph is None:
                with conv_c8.passes(plan[0]):
                    lp.unit(st, last=(iters == 1))   # eager once: packs weights, sizes the allocator
                done = 1
            if iters > done:
                lp.replay(plan[done:], st, capture_graph, last=True)     # (captures the kinds of unit it has not yet)
            # up-sampling mask head (core/update.py:107-110, :136) from the final hidden state: the final unit computes it
            # beside the flow head (loop_c8.C8Loop._mask)
            if lp.mask_out is not None:
                return lp.mask_out
            from . import conv_c8
            return .25 * conv2d(conv_c8.conv2d_c8([lp.hc8[0]], ub.mask[0], relu=True, cfg=1), ub.mask[2])

    def _iterate_graphed(self, fmap1, fmap2, net_list, inp_list, iters, flow_init, pending=None):
        """Same arithmetic as the eager loop; iterations 2..iters-1 are replays of one
        captured HIP graph (~60 launches per iteration leave the CPU out of the loop)."""
        args = self.args
        b, _, h, w = net_list[0].shape
        # The captured graph bakes in device pointers to the packed weight images, the merged z|r
        # weights and the biases, all of which are re-created when a parameter is replaced or written
        # (load_state_dict, .to(), optimiser steps) or the conv backend changes: those are part of the key.
        key = (fmap1.device, tuple(fmap1.shape), tuple(fmap2.shape), args.corr_implementation,
               self._weights_fingerprint(), self.rotate, self.pair_grus, self.pipeline_grus, self.fuse_lookup, self.use_c8)
        st = self._graph_state
        prebuilt, self._prebuilt = self._prebuilt, None
        static, self._static_ctx = self._static_ctx, None
        if st is None or st["key"] != key or prebuilt != (st["corr"], fmap1.data_ptr(), fmap2.data_ptr()):
            self._join_encoder(pending)      # (the volume is built here, on this stream, from the feature maps)
            pending = None
        if st is None or st["key"] != key:
            st = dict(key=key, graph=None, bound=None, fmap_shape=tuple(fmap1.shape))
            st["corr"] = CORR_IMPLEMENTATIONS[args.corr_implementation](
                fmap1, fmap2, radius=args.corr_radius, num_levels=args.corr_levels)
            st["coords0"] = coords_grid(b, h, w).to(fmap1.device)
            st["coords1"] = st["coords0"].clone()
            st["net"] = [t.clone() for t in net_list]
            # the three context terms of a scale are views of ONE buffer, the layout the context convolution writes:
            # the encoder pass of the following pairs writes them (and the hidden states) here itself (_context_targets)
            st["ctx"] = [torch.cat(list(scale), dim=1) for scale in inp_list]
            st["inp"] = [list(c.split(split_size=c.shape[1] // 3, dim=1)) for c in st["ctx"]]
            self._graph_state = st
        else:
            if prebuilt != (st["corr"], fmap1.data_ptr(), fmap2.data_ptr()):   # (else: encode() has rebuilt it for this pair)
                st["corr"].rebuild(fmap1, fmap2)
            st["coords1"].copy_(st["coords0"])
            if (self.adopt_encoder_outputs and static is not None and static[2] is net_list and static[3] is inp_list):
                ptrs = tuple(t.data_ptr() for t in net_list) + tuple(t.data_ptr() for sc in inp_list for t in sc)
                if st["bound"] != ptrs:
                    # the captured encoder pass's static outputs become the loop's state buffers: the loop updates the
                    # hidden states where the encoder writes them (every pair's encoder replay refreshes them) and reads
                    # the context terms in place; the captured units bake pointers in, so they are captured again
                    st["net"], st["inp"], st["bound"] = list(net_list), [list(sc) for sc in inp_list], ptrs
                    st["graph"] = None
                    if st.get("c8") is not None:
                        st["c8"].graph = st["c8"].graph_n = st["c8"].graph_last = None
            for dst, src in zip(st["net"], net_list):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            for ds, ss in zip(st["inp"], inp_list):
                for dst, src in zip(ds, ss):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
        if flow_init is not None:
            st["coords1"].add_(flow_init)
        if self.use_c8 and iters >= 3:
            from . import loop_c8
            if loop_c8.eligible(self, st["net"][0].shape):
                up_mask = self._iterate_c8(st, iters, pending)
                flow = st["coords1"] - st["coords0"]
                return flow, self.upsample_flow(flow, up_mask)[:, :1]
        self._join_encoder(pending)
        if self._can_pipeline() and self.rotate and self.pair_grus:
            up_mask = self._iterate_rotated(st, iters)
            flow = st["coords1"] - st["coords0"]
            return flow, self.upsample_flow(flow, up_mask)[:, :1]
        if self._can_pipeline():
            # prologue: gru32 of iteration 0 (the reference runs it first in every iteration)
            with harness(inplace_state=True):
                self.update_block(list(st["net"]), st["inp"], iter32=True, iter16=False, iter08=False, update=False)
            step = lambda mask: self._one_iteration_pipelined(st["corr"], st["coords0"], st["coords1"],  # noqa: E731
                                                              st["net"], st["inp"], mask, last=mask)
        else:
            step = lambda mask: self._one_iteration(st["corr"], st["coords0"], st["coords1"],  # noqa: E731
                                                    st["net"], st["inp"], mask)
        done = 0
        if st["graph"] is None:
            step(False)                      # eager once: packs weights, sizes the allocator
            done = 1
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with capture_graph(g):
                step(False)
            st["graph"] = g
        for _ in range(iters - 1 - done):
            replay_graph(st["graph"])
        up_mask = step(True)
        flow = st["coords1"] - st["coords0"]
        return flow, self.upsample_flow(flow, up_mask)[:, :1]

    def iterate(self, fmap1, fmap2, net_list, inp_list, iters, flow_init=None):
        """The hot path, raft_stereo.py:118-183 in test_mode: correlation build,
        then `iters` x (lookup, update block), then convex upsampling.  With `check_finite` (the default) the result's
        post-conditions are verified here, for every entry point: a flag time-out or activations that left the
        C8S scale window raise _RetryForward (a DktError) after the loop has fallen back / rescaled -- forward() repeats the
        pair, a direct caller calls encode() + iterate() again --, a non-finite result raises DktError."""
        st = self._graph_state
        if st is not None:
            st["c8_ran"] = False
        flow, flow_up = self._iterate(fmap1, fmap2, net_list, inp_list, iters, flow_init)
        if self.check_finite and flow_up.is_cuda:
            self._verify(flow_up)
        return flow, flow_up

    def _verify(self, flow_up):
        st = self._graph_state
        lp = st.get("c8") if (st is not None and st.get("c8_ran")) else None
        if lp is None:
            if not bool(torch.isfinite(flow_up).all()):
                raise _ffi.DktError(_NONFINITE)
            return
        s = lp.status(flow_up)               # one launch, one host synchronisation (csrc/status.hip)
        if s.err:
            # A fused ConvGRU (bit 0) or chain (bit 1) launch gave up waiting for a neighbour tile (csrc/gru_c8.hip: its
            # blocks were not all resident -- another process or model on this device): this result is wrong.  The word is
            # cleared, the loop leaves that form for good and the pair is computed again.
            if getattr(lp, "timed_out", 0) & s.err:
                raise _ffi.DktError("the refinement loop reported a flag time-out in a form that had already been switched off "
                                    "(csrc/gru_c8.hip, conv_c8.hip)")
            lp.timed_out = getattr(lp, "timed_out", 0) | s.err
            warnings.warn("dkt_stereo_amd: a %s launch timed out waiting for a neighbour tile; falling back to separate "
                          "launches for this model (one fused-GRU model per device)"
                          % ("fused ConvGRU" if s.err & 1 else "chain"))
            lp.on_error_word(s.err)
            raise _RetryForward("flag time-out")
        if lp.calibrated and not (s.finite and s.ranges_ok):
            # this pair's activations left the window the C8S scales were picked for: new scales, repeat the pair.  From the
            # maxima this pass left behind when they are all finite (no trial run); a trial run of the loop otherwise
            lp.recalibrations += 1
            if s.finite and all(math.isfinite(v) for v in s.maxima.reshape(-1).tolist()):
                lp.rescale_from(s.maxima)
                lp.calibrations += 1
            else:
                lp.calibrated = False
            raise _RetryForward("activation scales")
        if not s.finite:
            raise _ffi.DktError(_NONFINITE)

    def _iterate(self, fmap1, fmap2, net_list, inp_list, iters, flow_init=None):
        if self._mixed and fmap1.is_cuda and _conv.get_backend() == "f16x3" and not torch.is_grad_enabled():
            from . import loop_c8
            c8 = (self.use_c8 and self.use_hip_graph and iters >= 3 and not getattr(self, "_is_replica", False)
                  and self.args.corr_implementation == "reg" and loop_c8.eligible(self, net_list[0].shape))
            if not c8:                   # (the C8S loop takes its one-product schedule itself, see _iterate_c8)
                with _conv.use_backend("f16"):
                    return self._iterate_fp(fmap1, fmap2, net_list, inp_list, iters, flow_init)
        return self._iterate_fp(fmap1, fmap2, net_list, inp_list, iters, flow_init)

    def _join_encoder(self, pending):
        if pending is not None:
            torch.cuda.current_stream(pending.device).wait_stream(pending)

    def _iterate_fp(self, fmap1, fmap2, net_list, inp_list, iters, flow_init=None):
        args = self.args
        pending, self._pending_join = self._pending_join, None      # (the feature encoder's stream, not joined yet: see _encode)
        n = args.n_gru_layers
        # conv.calibrate() records activation ranges with a host synchronisation per layer: the plain loop (no stream capture,
        # every iteration observed) serves it
        # nn.DataParallel replicas are new modules with new parameter tensors (and new threads) on every forward: a captured
        # loop would be re-captured each time -- they run the plain loop; graph replay needs ONE persistent module per device
        if (self.use_hip_graph and not getattr(self, "_is_replica", False) and not _conv.calibrating() and iters >= 3 and fmap1.is_cuda and args.corr_implementation == "reg"
                and CORR_IMPLEMENTATIONS["reg"].__name__ == "CorrBlock1D"):
            return self._iterate_graphed(fmap1, fmap2, net_list, inp_list, iters, flow_init, pending)
        self._join_encoder(pending)
        self._prebuilt = self._static_ctx = None       # (the plain loop builds its own volume: nothing of encode()'s hand-over survives it)
        corr_block = CORR_IMPLEMENTATIONS[args.corr_implementation]
        corr_fn = corr_block(fmap1, fmap2, radius=args.corr_radius, num_levels=args.corr_levels)
        b, _, h, w = net_list[0].shape
        coords0 = coords_grid(b, h, w).to(fmap1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        net_list = list(net_list)
        up_mask = None
        for itr in range(iters):
            corr = self._lookup(corr_fn, coords1)
            flow = coords1 - coords0
            if n == 3 and args.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if n >= 2 and args.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=(n == 3), iter16=True, iter08=False, update=False)
            net_list, up_mask, delta_flow = self.update_block(
                net_list, inp_list, corr, flow, iter32=(n == 3), iter16=(n >= 2),
                need_mask=(itr == iters - 1))
            delta_flow[:, 1] = 0.0          # stereo: project onto the epipolar line
            coords1 = coords1 + delta_flow
        flow_up = self.upsample_flow(coords1 - coords0, up_mask)[:, :1]
        return coords1 - coords0, flow_up

    #: forward() verifies that its result is finite (one host sync per call).  The split-fp16
    #: convolutions turn an out-of-range activation (|x * 2^dkt_in_exp| >= 65520), Inf or NaN into a
    #: non-finite output instead of saturating it; this is where that becomes an error.
    check_finite = True

    def _forward_train(self, image1, image2, iters, flow_init):
        """raft_stereo.py:85-187 with test_mode=False: the up-sampled disparity of EVERY iteration (the sequence loss of
        tools/ft_dkt.py:223-242 weighs them), differentiable end to end.  The correlation block and the update operator run
        on this library's kernels as autograd nodes (corr._BuildFn / _LookupFn, conv.conv2d_autograd); the encoders' layers
        fall back to torch wherever their weights need gradients (extractor._Conv2d)."""
        args = self.args
        n = args.n_gru_layers
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        if self._two_encoders:
            cnet_list = self.cnet(image1, num_layers=n)
            fmap1, fmap2 = self.fnet([image1, image2])
        else:
            cnet_list, fmap1, fmap2 = self._variant_features(image1, image2, n)
        net_list = [torch.tanh(x[0]) for x in cnet_list]
        inp_list = [list(_conv.conv2d_autograd(torch.relu(x[1]), conv).split(split_size=conv.out_channels // 3, dim=1))
                    for x, conv in zip(cnet_list, self.context_zqr_convs)]
        corr_fn = CORR_IMPLEMENTATIONS[args.corr_implementation](fmap1.float(), fmap2.float(), radius=args.corr_radius,
                                                                 num_levels=args.corr_levels)
        b, _, h, w = net_list[0].shape
        coords0 = coords_grid(b, h, w).to(fmap1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        predictions = []
        for _ in range(iters):
            coords1 = coords1.detach()                                   # raft_stereo.py:153
            corr = corr_fn(coords1)
            flow = coords1 - coords0
            if n == 3 and args.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if n >= 2 and args.slow_fast_gru:
                net_list = self.update_block(net_list, inp_list, iter32=(n == 3), iter16=True, iter08=False, update=False)
            net_list, up_mask, delta_flow = self.update_block(net_list, inp_list, corr, flow, iter32=(n == 3), iter16=(n >= 2))
            # stereo: project the update onto the epipolar line (raft_stereo.py:165), without writing into an autograd output
            delta_flow = torch.cat([delta_flow[:, :1], torch.zeros_like(delta_flow[:, 1:])], dim=1)
            coords1 = coords1 + delta_flow
            predictions.append(self.upsample_flow(coords1 - coords0, up_mask)[:, :1])
        # the reference's return convention (raft_stereo.py:185-187): its loss reads results['disp_preds'] (loss.py:5,
        # tools/ft_dkt.py:218,262)
        return dict(disp_preds=predictions)

    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=False):
        if not test_mode:
            # the reference's default: {'disp_preds': every iteration's prediction} (differentiable when autograd is enabled;
            # under torch.no_grad() the same loop on the inference kernels, without the captured-graph fast path)
            with GPU_GUARD.shared():          # (kernel launches of this thread vs another thread's graph capture)
                out = self._forward_train(image1, image2, iters, flow_init)
                finite = (not self.check_finite) or bool(torch.isfinite(out["disp_preds"][-1]).all())
            if not finite:
                raise _ffi.DktError("RAFTStereo.forward(test_mode=False) produced non-finite disparities")
            return out
        if getattr(self, "_is_replica", False) and self.replica_shadows and image1.is_cuda:
            master = getattr(self, "_dp_master", lambda: None)()
            if master is not None:
                return _shadow_of(master, image1.device).run(master, image1, image2, iters=iters, flow_init=flow_init, test_mode=True)
        with torch.no_grad():
            return self._forward_test(image1, image2, iters, flow_init)

    def _forward_test(self, image1, image2, iters, flow_init):
        with GPU_GUARD.shared():                      # another thread's graph capture waits for this pass, and vice versa
            for _ in range(4):
                self._defer_join = True
                try:
                    fmap1, fmap2, net_list, inp_list = self.encode(image1, image2)
                finally:
                    self._defer_join = None
                try:
                    return self.iterate(fmap1, fmap2, net_list, inp_list, iters, flow_init)
                except _RetryForward:
                    # (the loop has changed form or scales; the encoder pass may have written the hidden states into the
                    # loop's own buffers, where the loop has updated them in place: produce them again)
                    continue
        raise _ffi.DktError("RAFTStereo.forward: the refinement loop asked for a fourth repeat of one pair "
                            "(flag time-outs / activation scales that do not settle)")
