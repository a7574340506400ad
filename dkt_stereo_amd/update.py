"""Update operator of RAFT-Stereo and IGEV-Stereo on MI355X.

Mirrors the module/parameter names of the reference's ``core/update.py``
(identical copy ``meta_arch/raft_stereo/update.py``) and
``meta_arch/igev_stereo/update.py`` so that ``state_dict()`` keys match and
published checkpoints load with ``strict=True``:

    RAFT  : encoder.{convc1,convc2,convf1,convf2,conv}, gru08/gru16/gru32.{convz,convr,convq},
            flow_head.{conv1,conv2}, mask.{0,2}
    IGEV  : encoder.{convc1,convc2,convd1,convd2,conv}, gru04/gru08/gru16, disp_head, mask_feat_4.0

What differs from the reference is how a GRU step is evaluated (core/update.py:23-32):
  * convz and convr run as ONE convolution with concatenated output channels;
  * default path: the gate arithmetic lives in the convolution epilogues
    (dkt_conv2d_f16s_gate_zr / _gate_out): two launches per GRU, no z|r / q pre-activation
    tensors, and the reference's torch.cat operands are read in place;
  * FUSE_GATES = False or the vendor-convolution backend: two streaming gate kernels
    (dkt_gru_gate_zr / _out) after the convolutions instead of ~12 elementwise launches.
Convolutions go through ``dkt_stereo_amd.conv.conv2d`` (see that module).
Inference only.
"""
import contextlib
import os
import threading
from types import SimpleNamespace

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ffi
from .conv import (_CACHE_LOCK, conv2d, conv2d_accumulate, conv2d_gate_out, conv2d_gate_out_pair, conv2d_gate_zr,
                   conv2d_autograd, conv2d_gate_zr_pair, conv2d_pair, few_eligible, get_backend, hip_eligible, pair_eligible)


class _Harness(threading.local):
    """Per-THREAD switches the loop harnesses (raft_stereo.RAFTStereo, igev_loop) set around their calls
    into the update block.  They used to be attributes toggled on the module, which two threads driving
    replicas of one module (nn.DataParallel, tools/ft_dkt.py:119) would have raced on."""
    inplace_state = False     # GRUs overwrite their hidden-state tensors instead of allocating new ones
    side_stream = None        # None: the module's default; False: the harness forks / joins itself
    before_fine = None        # hook called between the motion encoder and the finest GRU
    fine_interp = None        # callable returning interp(net[1], net[0]) computed elsewhere (another stream)
    pair_coarse = False       # run the coarsest GRU of the NEXT iteration in the launches of the finest GRU
    coarse_pool = None        # callable returning pool2x(net[1]) for that paired coarsest GRU
    branch_streams = False    # motion encoder's flow branch on its own stream
    accumulate_into = None    # tensor: the head adds its result to it in its epilogue; forward() returns delta = None


_HARNESS = _Harness()
class _CaptureGuard:
    """Keeps every other thread's GPU work of this library out of the way while one thread captures a HIP graph.

    A stream capture on ROCm 7.2 is invalidated by "unsafe" runtime calls of OTHER threads on the device -- a host
    synchronisation (``forward`` checks its result), a fresh hipMalloc of the caching allocator -- even in
    thread-local capture mode (observed: hipErrorStreamCaptureUnsupported in the bystander,
    hipErrorStreamCaptureInvalidated in the capturing thread).  The reference runs replicas from one Python
    thread per GPU (tools/ft_dkt.py:119), so: a forward pass holds the guard SHARED; a capture upgrades to
    EXCLUSIVE, i.e. waits until no other forward is in flight and keeps new ones out for the few milliseconds it
    takes.  Captures happen once per (thread, shape, weight set).  Re-entrant per thread."""

    def __init__(self):
        self._cv = threading.Condition()
        self._readers = 0
        self._writer = False
        self._waiting_writers = 0
        self._tls = threading.local()

    @contextlib.contextmanager
    def shared(self):
        depth = getattr(self._tls, "depth", 0)
        if depth == 0:
            with self._cv:
                while self._writer or self._waiting_writers:
                    self._cv.wait()
                self._readers += 1
        self._tls.depth = depth + 1
        try:
            yield
        finally:
            self._tls.depth = depth
            if depth == 0:
                with self._cv:
                    self._readers -= 1
                    self._cv.notify_all()

    @contextlib.contextmanager
    def exclusive(self):
        held = getattr(self._tls, "depth", 0) > 0        # this thread is inside a forward: give its slot back
        with self._cv:
            if held:
                self._readers -= 1
            self._waiting_writers += 1
            while self._writer or self._readers:
                self._cv.wait()
            self._waiting_writers -= 1
            self._writer = True
        try:
            yield
        finally:
            with self._cv:
                self._writer = False
                if held:
                    self._readers += 1
                self._cv.notify_all()


GPU_GUARD = _CaptureGuard()


@contextlib.contextmanager
def capture_graph(graph):
    with GPU_GUARD.exclusive():
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            yield


def replay_graph(graph):
    graph.replay()


@contextlib.contextmanager
def harness(**switches):
    """with harness(inplace_state=True, ...): thread-local, restored on exit."""
    prev = {k: getattr(_HARNESS, k) for k in switches}
    for k, v in switches.items():
        setattr(_HARNESS, k, v)
    try:
        yield _HARNESS
    finally:
        for k, v in prev.items():
            setattr(_HARNESS, k, v)


#: evaluate the GRU gates inside the convolution epilogues (dkt_conv2d_f16s_gate_zr/_out)
#: instead of the two streaming gate kernels
FUSE_GATES = True


#: the motion encoder's two independent 64 -> 64 layers (convc2, convf2 / convd2) share one launch
PAIR_ENCODER = True


def _batch_dense(t, hw):
    """True when each batch element of a (B,C,H,W) tensor is one dense C*H*W block."""
    return t.stride(3) == 1 and t.stride(2) == t.shape[3] and t.stride(1) == hw


class FlowHead(nn.Module):
    """core/update.py:6-14."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)

    def forward(self, x, outputs=None):
        """outputs=n: only the first n output channels of conv2 (the stereo callers drop the y component,
        raft_stereo.py:165: ``delta_flow[:,1] = 0.0`` -- not computing it halves the tail layer's work)."""
        head = self.conv2 if outputs is None or outputs >= self.conv2.out_channels else _leading_outputs(self.conv2, outputs)
        return conv2d(conv2d(x, self.conv1, relu=True), head)

    def add_to(self, x, target, outputs=None, diff=None):
        """target += head(x) in the tail layer's epilogue when it runs on the few-output kernel (else the plain add).
        `target`: (B, outputs, H, W) view of the running coordinates / disparity.  diff = (ref, dst): also
        dst = target_new - ref (the next iteration's flow = coords1 - coords0) in the same epilogue."""
        head = self.conv2 if outputs is None or outputs >= self.conv2.out_channels else _leading_outputs(self.conv2, outputs)
        hidden = conv2d(x, self.conv1, relu=True)
        if few_eligible(head) and target.is_cuda:
            return conv2d_accumulate(hidden, head, target, diff=diff)
        target.add_(conv2d(hidden, head))
        if diff is not None:
            torch.sub(target, diff[0], out=diff[1])
        return target


class _LayerView:
    """The first n output channels of a convolution layer (duck-types the `layer` argument of conv.conv2d)."""

    def __init__(self, layer, n):
        self._parent = layer
        self.weight = layer.weight[:n]
        self.bias = None if layer.bias is None else layer.bias[:n]
        self.padding, self.stride, self.dilation, self.groups = layer.padding, layer.stride, layer.dilation, layer.groups
        self.padding_mode = layer.padding_mode
        self._key = (layer.weight.data_ptr(), layer.weight._version,
                     None if layer.bias is None else (layer.bias.data_ptr(), layer.bias._version), n)

    @property
    def dkt_in_exp(self):                      # the activation exponent (conv.calibrate) lives on the real layer
        return getattr(self._parent, "dkt_in_exp", 0)

    @dkt_in_exp.setter
    def dkt_in_exp(self, e):
        self._parent.dkt_in_exp = e


class _ScaledLayer(_LayerView):
    """`layer` with weight and bias multiplied by a power of two: scale * layer(x) in the layer's own launch (exact in fp32,
    so bit-identical to scaling the result -- the reference's `.25 * self.mask(net)`, core/update.py:136)."""

    def __init__(self, layer, scale):
        super().__init__(layer, layer.weight.shape[0])
        with torch.no_grad():
            self.weight = (layer.weight.detach() * scale).contiguous()
            self.bias = None if layer.bias is None else (layer.bias.detach() * scale).contiguous()
        self._key = self._key[:3] + (("scale", scale),)


def _scaled_layer(layer, scale):
    m, e = math.frexp(scale)
    if m != 0.5:
        raise ValueError("_scaled_layer: %r is not a power of two" % (scale,))
    key = (layer.weight.data_ptr(), layer.weight._version,
           None if layer.bias is None else (layer.bias.data_ptr(), layer.bias._version), ("scale", scale))
    view = layer.__dict__.get("_dkt_scaled")
    if view is None or view._key != key:
        view = layer.__dict__["_dkt_scaled"] = _ScaledLayer(layer, scale)
    return view


def _leading_outputs(layer, n):
    key = (layer.weight.data_ptr(), layer.weight._version,
           None if layer.bias is None else (layer.bias.data_ptr(), layer.bias._version), n)
    view = layer.__dict__.get("_dkt_view")
    if view is None or view._key != key:
        view = layer.__dict__["_dkt_view"] = _LayerView(layer, n)
    return view


class DispHead(FlowHead):
    """meta_arch/igev_stereo/update.py:16-24."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=1):
        super().__init__(input_dim, hidden_dim, output_dim)


class _MergedZR:
    """convz | convr as one Conv2d-like layer.  The activation exponent (conv.calibrate) is kept on the real convz module:
    it survives a rebuild of this object and is seen by the loop's weight fingerprint (RAFTStereo._weights_fingerprint)."""

    def __init__(self, convz, weight, bias):
        self._convz = convz
        self.weight, self.bias = weight, bias
        self.padding, self.stride, self.dilation, self.groups = convz.padding, convz.stride, convz.dilation, convz.groups
        self.padding_mode = convz.padding_mode

    @property
    def dkt_in_exp(self):
        return getattr(self._convz, "dkt_in_exp", 0)

    @dkt_in_exp.setter
    def dkt_in_exp(self, e):
        self._convz.dkt_in_exp = e


class ConvGRU(nn.Module):
    """core/update.py:16-32 == meta_arch/igev_stereo/update.py:26-41."""

    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__()
        pad = kernel_size // 2
        self.convz = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convr = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self.convq = nn.Conv2d(hidden_dim + input_dim, hidden_dim, kernel_size, padding=pad)
        self._zr_cache = None        # {device: (key, merged layer)}; shared by replicate()'s shallow copies

    def _merged_zr(self):
        """convz | convr as one Conv2d-like (weight, bias) pair per device, rebuilt whenever
        either parameter tensor is replaced or written (load_state_dict, .to())."""
        wz, wr = self.convz.weight, self.convr.weight
        key = (wz.data_ptr(), wr.data_ptr(), wz._version, wr._version,
               self.convz.bias.data_ptr(), self.convr.bias.data_ptr(),
               self.convz.bias._version, self.convr.bias._version)
        with _CACHE_LOCK:
            cache = self._zr_cache
            if cache is None:
                cache = self._zr_cache = {}
            hit = cache.get(str(wz.device))
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    zr = _MergedZR(self.convz, torch.cat([wz, wr], dim=0).contiguous(),
                                   torch.cat([self.convz.bias, self.convr.bias], dim=0).contiguous())
                hit = cache[str(wz.device)] = (key, zr)
            return hit[1]

    def forward(self, h, cz, cr, cq, *x_list, out=None):
        """`out` (optional, may be `h`): where the new hidden state is written."""
        _ffi.require_gpu(h, cz, cr, cq, *x_list)
        _ffi.require_no_grad(h, cz, cr, cq, *x_list)
        B, Ch, H, W = h.shape
        HW = H * W
        L = _ffi.lib()
        dev, st = _ffi.device_of(h), _ffi.stream_of(h)
        if not _batch_dense(h, HW):
            h = h.contiguous()
        cz, cr, cq = [t if _batch_dense(t, HW) else t.contiguous() for t in (cz, cr, cq)]

        if FUSE_GATES and hip_eligible(self.convq) and Ch % 64 == 0:
            # gates live in the convolution epilogues: two launches per GRU, no z|r / q
            # pre-activation tensors, no torch.cat (operands are read in place)
            z, rh = conv2d_gate_zr([h, *x_list], self._merged_zr(), cz, cr, h)
            return conv2d_gate_out([rh, *x_list], self.convq, cq, z, h, out=out)

        if get_backend() == "miopen":
            # vendor convolutions want one dense operand: build [h | x] once and
            # let the gate kernel overwrite its first Ch channels with r*h
            hx = torch.cat([h, *x_list], dim=1)
            zr_in, rh, q_in = hx, hx, hx
        else:
            # own convolution kernel reads the cat operands in place: no copies
            rh = torch.empty((B, Ch, H, W), device=h.device, dtype=torch.float32)
            zr_in, q_in = [h, *x_list], [rh, *x_list]
        azr = conv2d(zr_in, self._merged_zr())                    # (B, 2Ch, H, W)
        _ffi.require_gpu(azr)        # fp32 only: under torch.autocast the vendor conv returns half
        z = torch.empty((B, Ch, H, W), device=h.device, dtype=torch.float32)
        # z = sigmoid(az+cz); r = sigmoid(ar+cr); rh <- r*h
        rc = L.dkt_gru_gate_zr(azr.data_ptr(), cz.data_ptr(), cz.stride(0), cr.data_ptr(), cr.stride(0),
                               h.data_ptr(), h.stride(0), z.data_ptr(), rh.data_ptr(), rh.stride(0),
                               B, Ch, HW, dev, st)
        _ffi.check(rc, "dkt_gru_gate_zr")
        aq = conv2d(q_in, self.convq)
        _ffi.require_gpu(aq)
        if out is None:
            out = torch.empty((B, Ch, H, W), device=h.device, dtype=torch.float32)
        # q = tanh(aq+cq); h' = (1-z)*h + z*q
        rc = L.dkt_gru_gate_out(aq.data_ptr(), cq.data_ptr(), cq.stride(0), z.data_ptr(),
                                h.data_ptr(), h.stride(0), out.data_ptr(), out.stride(0),
                                B, Ch, HW, dev, st)
        _ffi.check(rc, "dkt_gru_gate_out")
        return out


def gru_pair(gru_a, args_a, gru_b, args_b):
    """Two independent ConvGRU steps (core/update.py:23-32 twice) in TWO launches instead of four:
    the z|r convolutions of both share one launch, so do the q convolutions (dkt_conv2d_f16s_pair).
    args = (h, cz, cr, cq, x_list, out).  Meant for a large and a small image: the small one's tiles
    fit in the tile-quantisation slack of the large one's launch.  Falls back to two plain calls
    when the fused-gate path or the pairing does not apply.  Returns (h_a', h_b')."""
    ok = (FUSE_GATES and hip_eligible(gru_a.convq) and hip_eligible(gru_b.convq)
          and pair_eligible(gru_a.convq, gru_b.convq) and pair_eligible(gru_a.convz, gru_b.convz)
          and args_a[0].shape[1] % 64 == 0 and args_b[0].shape[1] % 64 == 0
          and args_a[0].shape[1] == args_b[0].shape[1])
    if not ok:
        ha, cza, cra, cqa, xa, oa = args_a
        hb, czb, crb, cqb, xb, ob = args_b
        return gru_a(ha, cza, cra, cqa, *xa, out=oa), gru_b(hb, czb, crb, cqb, *xb, out=ob)
    prep = []
    for gru, (h, cz, cr, cq, xs, out) in ((gru_a, args_a), (gru_b, args_b)):
        _ffi.require_gpu(h, cz, cr, cq, *xs)
        _ffi.require_no_grad(h, cz, cr, cq, *xs)
        HW = h.shape[2] * h.shape[3]
        if not _batch_dense(h, HW):
            h = h.contiguous()
        cz, cr, cq = [t if _batch_dense(t, HW) else t.contiguous() for t in (cz, cr, cq)]
        prep.append((gru, h, cz, cr, cq, list(xs), out))
    (za, rha), (zb, rhb) = conv2d_gate_zr_pair(*[([h, *xs], g._merged_zr(), cz, cr, h) for g, h, cz, cr, cq, xs, out in prep])
    (ga, ha, _, _, cqa, xsa, oa), (gb, hb, _, _, cqb, xsb, ob) = prep
    return tuple(conv2d_gate_out_pair(([rha, *xsa], ga.convq, cqa, za, ha, oa), ([rhb, *xsb], gb.convq, cqb, zb, hb, ob)))


class BasicMotionEncoder(nn.Module):
    """core/update.py:64-85 (RAFT).  ``cor_planes = corr_levels * (2*corr_radius+1)``."""

    _branch = ("convf1", "convf2")
    _cor_mult = 1
    _aux_ch = 2

    def __init__(self, args):
        super().__init__()
        self.args = args
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) * self._cor_mult
        self.convc1 = nn.Conv2d(cor_planes, 64, 1, padding=0)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        setattr(self, self._branch[0], nn.Conv2d(self._aux_ch, 64, 7, padding=3))
        setattr(self, self._branch[1], nn.Conv2d(64, 64, 3, padding=1))
        self.conv = nn.Conv2d(64 + 64, 128 - self._aux_ch, 3, padding=1)

    # harness(branch_streams=True): the flow branch runs on its own stream beside the correlation
    # branch.  Only set when the encoder itself runs on the capture's origin stream (a fork nested
    # inside another forked stream crashed hipStreamEndCapture on ROCm 7.2).

    def _cor1(self, corr):
        """relu(convc1(corr)).  `corr` may be a corr.DeferredLookup (the loop harnesses pass one): lookup and
        this 1x1 layer then run as ONE kernel on the exact-fp32 matrix pipe (dkt_corr1d_lookup_conv1x1)."""
        if hasattr(corr, "materialize"):
            out = corr.conv1x1(self.convc1, relu=True) if get_backend() != "miopen" else None
            if out is not None:
                return out
            corr = corr.materialize()
        return conv2d(corr, self.convc1, relu=True)

    def forward(self, flow, corr):
        if _HARNESS.branch_streams and flow.is_cuda:
            cur = torch.cuda.current_stream(flow.device)
            aux = _side_stream(flow.device, slot=1)
            aux.wait_stream(cur)
            with torch.cuda.stream(aux):
                flo = conv2d(conv2d(flow, getattr(self, self._branch[0]), relu=True),
                             getattr(self, self._branch[1]), relu=True)
            cor = conv2d(self._cor1(corr), self.convc2, relu=True)
            cur.wait_stream(aux)
        else:
            cor1 = self._cor1(corr)
            flo1 = conv2d(flow, getattr(self, self._branch[0]), relu=True)
            second = getattr(self, self._branch[1])
            if PAIR_ENCODER and flow.is_cuda and pair_eligible(self.convc2, second):
                # the two branches' 3x3 layers (64 -> 64 each) are independent: one launch
                cor, flo = conv2d_pair((cor1, self.convc2, True), (flo1, second, True))
            else:
                cor = conv2d(cor1, self.convc2, relu=True)
                flo = conv2d(flo1, second, relu=True)
        # [conv output (126/127 ch) | flow]: the convolution writes straight into the first
        # channels of the 128-channel motion-feature buffer (no torch.cat of the big part)
        B, _, H, W = flow.shape
        nout = self.conv.weight.shape[0]
        feat = self.feature_buffer_of(flow)
        if feat is None:
            feat = torch.empty((B, nout + flow.shape[1], H, W), device=flow.device, dtype=torch.float32)
            feat[:, nout:] = flow
        conv2d([cor, flo], self.conv, relu=True, out=feat[:, :nout])
        return feat

    def new_feature_buffer(self, B, H, W, device):
        """(feat, flow_view): the 128-channel motion-feature buffer and its trailing flow / disparity channels.  A
        caller that computes the flow straight into ``flow_view`` (torch.sub(..., out=flow_view)) and passes that view
        as ``flow`` saves the copy of core/update.py:85's torch.cat."""
        nout = self.conv.weight.shape[0]
        feat = torch.empty((B, nout + self._aux_ch, H, W), device=device, dtype=torch.float32)
        view = feat[:, nout:]
        view._dkt_feat = feat
        return feat, view

    def feature_buffer_of(self, flow):
        feat = getattr(flow, "_dkt_feat", None)
        if (feat is not None and feat.shape[1] == self.conv.weight.shape[0] + flow.shape[1]
                and flow.data_ptr() == feat[:, self.conv.weight.shape[0]:].data_ptr() and feat.shape[2:] == flow.shape[2:]):
            return feat
        return None


class BasicMotionEncoderIGEV(BasicMotionEncoder):
    """meta_arch/igev_stereo/update.py:73-92: cor_planes = L*(2r+1)*(8+1), a
    1-channel disparity branch (convd1/convd2) and a 127-channel output conv."""

    _branch = ("convd1", "convd2")
    _cor_mult = 9
    _aux_ch = 1


def pool2x(x):
    """core/update.py:87-88: F.avg_pool2d(x, 3, stride=2, padding=1) (dkt_pool2x)."""
    _ffi.require_gpu(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
    rc = _ffi.lib().dkt_pool2x(x.data_ptr(), y.data_ptr(), B * C, H, W, _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_pool2x")
    return y


def interp(x, dest):
    """core/update.py:93-95: F.interpolate(x, dest.shape[2:], mode='bilinear',
    align_corners=True) (dkt_interp_bilinear)."""
    _ffi.require_gpu(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    Ho, Wo = dest.shape[2:]
    y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32)
    rc = _ffi.lib().dkt_interp_bilinear(x.data_ptr(), y.data_ptr(), B * C, H, W, Ho, Wo,
                                        _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_interp_bilinear")
    return y


class _SideStreams(threading.local):
    def __init__(self):
        self.streams = {}


_SIDE_STREAMS = _SideStreams()


def _side_stream(device, slot=0):
    """Auxiliary HIP streams per (thread, device) (slot 0: coarse GRUs / motion encoder / fnet; slot 1:
    the flow branch of the motion encoder).  Per thread, so that two threads running (and capturing)
    forwards on one device never fork into the same stream."""
    idx = torch.device(device).index
    key = (idx if idx is not None else torch.cuda.current_device(), slot)
    st = _SIDE_STREAMS.streams.get(key)
    if st is None:
        st = _SIDE_STREAMS.streams[key] = torch.cuda.Stream(device=key[0])
    return st


class BasicMultiUpdateBlock(nn.Module):
    """core/update.py:97-138 (RAFT-Stereo).  ``forward`` keeps the reference's
    signature and, like it, updates the caller's ``net`` list in place.
    Extra keyword ``need_mask`` (default True = reference behaviour): the harness
    passes False on all but the last iteration of ``test_mode`` inference, where
    the reference computes the up-sampling mask only to discard it
    (raft_stereo.py:170-171)."""

    def __init__(self, args, hidden_dims=[]):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        encoder_output_dim = 128
        n = args.n_gru_layers
        self.gru08 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (n > 1))
        self.gru16 = ConvGRU(hidden_dims[1], hidden_dims[0] * (n == 3) + hidden_dims[2])
        self.gru32 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHead(hidden_dims[2], hidden_dim=256, output_dim=2)
        factor = 2 ** self.args.n_downsample
        self.mask = nn.Sequential(
            nn.Conv2d(hidden_dims[2], 256, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(256, (factor ** 2) * 9, 1, padding=0))

    #: run the motion encoder concurrently with gru32/gru16 on a second stream (attribute side_stream);
    #: a harness that forks / joins itself overrides it per thread with harness(side_stream=False)
    side_stream = True
    # Other per-thread harness switches (see _Harness): inplace_state -- GRUs overwrite their hidden-state
    # tensors (the reference rebinds net[i] to a fresh tensor; nothing else may hold the old one);
    # before_fine -- hook between the motion encoder and the finest GRU.

    def _gru_stack(self, net, inp, fine, mid, coarse, motion, it_fine, it_mid, it_coarse):
        n = self.args.n_gru_layers
        hs = _HARNESS
        o = (lambda t: t) if hs.inplace_state else (lambda t: None)
        use_side = self.side_stream if hs.side_stream is None else hs.side_stream
        # The motion encoder (5 convolutions of the finest scale) does not depend on the two
        # coarse GRUs, whose small images leave most CUs idle (gru16: 115 tiles, gru32: 69, for
        # 512 resident blocks): it runs on a second HIP stream beside them and joins before
        # gru08.  Inside the captured iteration graph this is a parallel branch.
        side = main = None
        if it_fine and (it_coarse or it_mid) and use_side and net[0].is_cuda:
            main = torch.cuda.current_stream(net[0].device)
            side = _side_stream(net[0].device)
            side.wait_stream(main)              # fork
            with torch.cuda.stream(side):
                mf = motion()
        if it_coarse:
            net[2] = coarse(net[2], *(inp[2]), pool2x(net[1]), out=o(net[2]))
        if it_mid:
            if n > 2:
                net[1] = mid(net[1], *(inp[1]), pool2x(net[0]), interp(net[2], net[1]), out=o(net[1]))
            else:
                net[1] = mid(net[1], *(inp[1]), pool2x(net[0]), out=o(net[1]))
        if it_fine:
            if side is not None:
                main.wait_stream(side)          # join: gru08 consumes the motion features
            else:
                mf = motion()
            if hs.before_fine is not None:
                hs.before_fine()                # harness hook: wait for a coarse GRU running on another stream
            if n > 1:
                up = hs.fine_interp() if hs.fine_interp is not None else interp(net[1], net[0])
                if hs.pair_coarse and n == 3:
                    # pipelined harness: net[1] already holds this iteration's middle GRU, so the coarsest GRU
                    # of the next iteration (inputs net[2], pool2x(net[1]); core/update.py:118-119) is ready
                    # to run and independent of the finest GRU: both share their two launches
                    net[0], net[2] = gru_pair(fine, (net[0], *(inp[0]), [mf, up], o(net[0])),
                                              coarse, (net[2], *(inp[2]), [hs.coarse_pool() if hs.coarse_pool is not None
                                                                            else pool2x(net[1])], o(net[2])))
                else:
                    net[0] = fine(net[0], *(inp[0]), mf, up, out=o(net[0]))
            else:
                net[0] = fine(net[0], *(inp[0]), mf, out=o(net[0]))
        return net

    # ---- autograd path (training through the update operator, tools/ft_dkt.py:223-242; VERDICT r02 missing #2) -------
    def _wants_grad(self, net, inp, corr, aux):
        if not torch.is_grad_enabled():
            return False
        ts = [t for t in list(net) + [t for scale in inp for t in scale] + [corr, aux] if torch.is_tensor(t)]
        return any(t.requires_grad for t in ts) or any(p.requires_grad for p in self.parameters())

    @staticmethod
    def _gru_autograd(gru, h, cz, cr, cq, *xs):
        """core/update.py:23-32 on differentiable convolutions (conv.conv2d_autograd); z and r share one convolution."""
        hx = torch.cat([h, *xs], dim=1)
        zr = SimpleNamespace(weight=torch.cat([gru.convz.weight, gru.convr.weight], 0),
                             bias=torch.cat([gru.convz.bias, gru.convr.bias], 0), padding=gru.convz.padding)
        a = conv2d_autograd(hx, zr)
        ch = h.shape[1]
        z = torch.sigmoid(a[:, :ch] + cz)
        r = torch.sigmoid(a[:, ch:] + cr)
        q = torch.tanh(conv2d_autograd(torch.cat([r * h, *xs], dim=1), gru.convq) + cq)
        return (1 - z) * h + z * q

    def _encoder_autograd(self, aux, corr):
        e = self.encoder
        if hasattr(corr, "materialize"):
            corr = corr.materialize()
        cor = conv2d_autograd(conv2d_autograd(corr, e.convc1, relu=True), e.convc2, relu=True)
        b1, b2 = (getattr(e, n) for n in e._branch)
        aux_f = conv2d_autograd(conv2d_autograd(aux, b1, relu=True), b2, relu=True)
        return torch.cat([conv2d_autograd([cor, aux_f], e.conv, relu=True), aux], dim=1)

    def _stack_autograd(self, net, inp, grus, aux, corr, flags):
        """core/update.py:115-133 (IGEV: igev_stereo/update.py:122-135): coarse to fine, every operator an autograd node."""
        n = self.args.n_gru_layers
        fine, mid, coarse = grus
        it_fine, it_mid, it_coarse = flags
        net = list(net)
        p2 = lambda t: F.avg_pool2d(t, 3, stride=2, padding=1)
        up = lambda t, dest: F.interpolate(t, dest.shape[2:], mode="bilinear", align_corners=True)
        if it_coarse:
            net[2] = self._gru_autograd(coarse, net[2], *inp[2], p2(net[1]))
        if it_mid:
            xs = [p2(net[0])] + ([up(net[2], net[1])] if n > 2 else [])
            net[1] = self._gru_autograd(mid, net[1], *inp[1], *xs)
        if it_fine:
            mf = self._encoder_autograd(aux, corr)
            xs = [mf] + ([up(net[1], net[0])] if n > 1 else [])
            net[0] = self._gru_autograd(fine, net[0], *inp[0], *xs)
        return net

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True,
                update=True, need_mask=True):
        if self._wants_grad(net, inp, corr, flow):
            net = self._stack_autograd(net, inp, (self.gru08, self.gru16, self.gru32), flow, corr, (iter08, iter16, iter32))
            if not update:
                return net
            fh = self.flow_head
            delta_flow = conv2d_autograd(conv2d_autograd(net[0], fh.conv1, relu=True), fh.conv2)
            mask = .25 * conv2d_autograd(conv2d_autograd(net[0], self.mask[0], relu=True), self.mask[2]) if need_mask else None
            return net, mask, delta_flow
        net = self._gru_stack(net, inp, self.gru08, self.gru16, self.gru32,
                              lambda: self.encoder(flow, corr), iter08, iter16, iter32)
        if not update:
            return net
        delta_flow = self.flow_head(net[0])
        mask = None
        if need_mask:
            # scale mask to balance gradients (core/update.py:137)
            mask = .25 * conv2d(conv2d(net[0], self.mask[0], relu=True), self.mask[2])
        return net, mask, delta_flow


class BasicMultiUpdateBlockIGEV(BasicMultiUpdateBlock):
    """meta_arch/igev_stereo/update.py:104-142: GRUs named gru04/gru08/gru16,
    ``disp_head`` (1 channel) and ``mask_feat_4`` = conv3x3(128->32)+ReLU.
    forward(net, inp, corr, disp, iter04, iter08, iter16, update) ->
    (net, mask_feat_4, delta_disp)."""

    def __init__(self, args, hidden_dims=[]):
        nn.Module.__init__(self)
        self.args = args
        self.encoder = BasicMotionEncoderIGEV(args)
        encoder_output_dim = 128
        n = args.n_gru_layers
        self.gru04 = ConvGRU(hidden_dims[2], encoder_output_dim + hidden_dims[1] * (n > 1))
        self.gru08 = ConvGRU(hidden_dims[1], hidden_dims[0] * (n == 3) + hidden_dims[2])
        self.gru16 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.disp_head = DispHead(hidden_dims[2], hidden_dim=256, output_dim=1)
        self.mask_feat_4 = nn.Sequential(
            nn.Conv2d(hidden_dims[2], 32, 3, padding=1),
            nn.ReLU(inplace=True))

    def _replicate_for_data_parallel(self):
        """nn.DataParallel's per-forward copy knows the block it was made from: igev_loop.igev_iterate hands a replica's loop
        to a persistent per-device copy of THAT block (igev_loop._ShadowBlock), as raft_stereo.RAFTStereo does for itself."""
        import weakref
        replica = super()._replicate_for_data_parallel()
        replica._dp_master = weakref.ref(self)
        return replica

    def forward(self, net, inp, corr=None, disp=None, iter04=True, iter08=True, iter16=True,
                update=True, need_mask=True):
        if self._wants_grad(net, inp, corr, disp):
            net = self._stack_autograd(net, inp, (self.gru04, self.gru08, self.gru16), disp, corr, (iter04, iter08, iter16))
            if not update:
                return net
            dh = self.disp_head
            delta_disp = conv2d_autograd(conv2d_autograd(net[0], dh.conv1, relu=True), dh.conv2)
            mask_feat_4 = conv2d_autograd(net[0], self.mask_feat_4[0], relu=True) if need_mask else None
            return net, mask_feat_4, delta_disp
        net = self._gru_stack(net, inp, self.gru04, self.gru08, self.gru16,
                              lambda: self.encoder(disp, corr), iter04, iter08, iter16)
        if not update:
            return net
        if _HARNESS.accumulate_into is not None:
            self.disp_head.add_to(net[0], _HARNESS.accumulate_into)      # disp += delta in the tail layer's epilogue
            delta_disp = None
        else:
            delta_disp = self.disp_head(net[0])
        mask_feat_4 = conv2d(net[0], self.mask_feat_4[0], relu=True) if need_mask else None
        return net, mask_feat_4, delta_disp
