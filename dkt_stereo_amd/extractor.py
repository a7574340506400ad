"""Feature / context encoders: the callers' side of the hot path.

They run once per pair and produce the feature maps and GRU context the hot path consumes;
published checkpoints must load, so parameter names follow the reference's
``core/extractor.py`` (``BasicEncoder`` :122-197, ``MultiBasicEncoder`` :199-300,
``ResidualBlock`` :6-60).  On a HIP device in inference every convolution runs on this
library's kernels (1x1 / 3x3 with stride 1 or 2: dkt_conv2d_f16s[_strided]; the 7x7 stem:
dkt_conv2d_stem7), eval-mode BatchNorm is folded into the convolution, instance norm and the
residual join are streaming kernels; anything else (training, CPU, group norm) is plain torch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ffi
import os

from .conv import (_CACHE_LOCK, conv2d, conv2d_fused, conv2d_fused_pair, conv2d_pair, conv2d_stats, fused_eligible,
                   pair_eligible, stats_eligible)

#: DKT_FUSE_ENCODER=0: separate normalise / residual-join passes around the encoders' convolutions (A/B switch)
FUSE_ENCODER = os.environ.get("DKT_FUSE_ENCODER", "1") != "0"
#: DKT_EPILOGUE_STATS=0: instance-norm statistics by a pass of their own (dkt_instance_norm_stats) instead of in the producing
#: convolution's epilogue
EPILOGUE_STATS = os.environ.get("DKT_EPILOGUE_STATS", "1") != "0"
#: DKT_CNET_STREAMS=0: the context encoder's output heads one after the other on the trunk's stream
CNET_STREAMS = os.environ.get("DKT_CNET_STREAMS", "1") != "0"


#: DKT_C8_ENCODER=1: the full-resolution stage (conv1 + layer1) on conv_c8.  Opt-in: the 64 -> 64 convolutions themselves
#: drop from 270-300 to 205-215 us per image at 736 x 1248, but the fp32 residual traffic of the join epilogue (batch-norm
#: encoder) and the normalise-to-C8S passes (instance-norm encoder) give the gain back: 6.2 -> 6.5 ms fnet, 2.5 -> 2.6 ms
#: cnet (profiles/r03_encoder_c8.txt)
C8_ENCODER = os.environ.get("DKT_C8_ENCODER", "0") == "1"
#: full-resolution pixels from which layer1 takes the C8S path (its tiles are 8 rows x 32 columns; smaller images leave CUs idle)
C8_ENCODER_MIN_PIXELS = 100000
#: tile shape of the 64 -> 64 layers (conv_c8.hip c8_dispatch)
C8_ENCODER_CFG = 3


def _hip_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad)


def norm_act(norm, x, relu):
    """norm(x) [+ ReLU].  InstanceNorm2d (affine-free, instance statistics: fnet) runs as the
    fused dkt_instance_norm; every other norm is torch (+ F.relu)."""
    if (isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats and _hip_ok(x)):
        x = x.contiguous()
        n, c, h, w = x.shape
        L = _ffi.lib()
        ws = torch.empty(L.dkt_instance_norm_workspace(n * c, h * w), device=x.device, dtype=torch.uint8)
        y = torch.empty_like(x)
        rc = L.dkt_instance_norm(x.data_ptr(), y.data_ptr(), ws.data_ptr(), n * c, h * w, float(norm.eps),
                                 int(relu), _ffi.device_of(x), _ffi.stream_of(x))
        _ffi.check(rc, "dkt_instance_norm")
        return y
    y = norm(x)
    return F.relu(y) if relu else y


class _Folded:
    """A convolution with an eval-mode BatchNorm folded into its weights (duck-types the layer
    argument of conv.conv2d)."""


def _tensor_key(t):
    return None if t is None else (t.data_ptr(), t._version)


def _folded(conv, bn):
    """conv followed by bn (running statistics):  bn(conv(x)) = conv'(x) with
    w' = w * g, b' = (b - mean) * g + beta, g = gamma / sqrt(var + eps)  (per output channel).
    Cached on the conv module; rebuilt when any of the tensors is replaced or written."""
    with _CACHE_LOCK:
        return _folded_locked(conv, bn)


def _folded_locked(conv, bn):
    key = tuple(_tensor_key(t) for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cache = conv.__dict__.setdefault("_dkt_folded", {})      # per device: replicas share this dict
    slot = str(conv.weight.device)
    hit = cache.get(slot)
    if hit is not None and hit.key == key:
        return hit
    with torch.no_grad():
        g = torch.rsqrt(bn.running_var.double() + bn.eps)
        if bn.weight is not None:
            g = g * bn.weight.double()
        b = conv.bias.double() if conv.bias is not None else torch.zeros_like(g)
        b = (b - bn.running_mean.double()) * g
        if bn.bias is not None:
            b = b + bn.bias.double()
        f = _Folded()
        f.weight = (conv.weight.double() * g.view(-1, 1, 1, 1)).float().contiguous()
        f.bias = b.float().contiguous()
    f.padding = conv.padding
    f.stride = conv.stride
    f.dilation = conv.dilation        # conv2d sends dilated / grouped layers to the vendor library
    f.groups = conv.groups
    f.key = key
    cache[slot] = f
    return f


def _foldable(conv, norm, x):
    return (isinstance(norm, nn.BatchNorm2d) and not norm.training and norm.track_running_stats
            and isinstance(conv, nn.Conv2d) and _hip_ok(x) and conv.padding_mode == 'zeros'
            and not (torch.is_grad_enabled() and conv.weight.requires_grad))


def conv_norm_act(conv, norm, x, relu):
    """norm(conv(x)) [+ ReLU].  An eval-mode BatchNorm (cnet: frozen statistics,
    raft_stereo.py:56-59) is folded into the convolution and the ReLU into its epilogue, which
    removes two full passes over the activation per layer; other norms go through norm_act."""
    if _foldable(conv, norm, x):
        return conv2d(x, _folded(conv, norm), relu=relu)      # conv2d honours f.stride
    return norm_act(norm, conv(x), relu)


def instance_norm_params(norm, x, stats=None):
    """(N*C, 2) float tensor of (mean, 1/sqrt(var + eps)) per plane of x (dkt_instance_norm_stats +
    dkt_instance_norm_finalize): the `in_norm` operand of conv.conv2d_fused.  `stats` (conv.OutStats): the partial sums the
    convolution that produced x left behind -- the statistics pass is skipped."""
    x = x if x.is_contiguous() else x.contiguous()
    n, c, h, w = x.shape
    L = _ffi.lib()
    out = torch.empty((n * c, 2), device=x.device, dtype=torch.float32)
    if stats is not None:
        ws = stats.part
    else:
        ws = torch.empty(L.dkt_instance_norm_workspace(n * c, h * w), device=x.device, dtype=torch.uint8)
        rc = L.dkt_instance_norm_stats(x.data_ptr(), ws.data_ptr(), n * c, h * w, _ffi.device_of(x), _ffi.stream_of(x))
        _ffi.check(rc, "dkt_instance_norm_stats")
    rc = L.dkt_instance_norm_finalize(ws.data_ptr(), n * c, h * w, float(norm.eps), out.data_ptr(),
                                      _ffi.device_of(x), _ffi.stream_of(x))
    _ffi.check(rc, "dkt_instance_norm_finalize")
    return out


def _plain_instance_norm(norm):
    return isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats


def norm_add_relu(norm, x, c, c_stats=None):
    """relu(x + relu(norm(c))): the tail of a residual block.  With an affine-free instance norm
    (fnet) the normalisation of c is folded into the join (dkt_instance_norm_stats +
    dkt_instance_norm_add_relu: one pass over c for the statistics -- none when the producing convolution left them in
    `c_stats` --, one fused pass), otherwise norm_act + add_relu."""
    lazy = x if isinstance(x, LazyNorm) else None
    if lazy is not None:
        x = lazy.raw
    if _plain_instance_norm(norm) and _hip_ok(x) and _hip_ok(c) and x.shape == c.shape:
        x = x.contiguous()
        c = c.contiguous()
        n, ch, h, w = c.shape
        L = _ffi.lib()
        y = torch.empty_like(c)
        if c_stats is not None:
            ws = c_stats.part
        else:
            ws = torch.empty(L.dkt_instance_norm_workspace(n * ch, h * w), device=c.device, dtype=torch.uint8)
            rc = L.dkt_instance_norm_stats(c.data_ptr(), ws.data_ptr(), n * ch, h * w, _ffi.device_of(c), _ffi.stream_of(c))
            _ffi.check(rc, "dkt_instance_norm_stats")
        if lazy is not None:
            rc = L.dkt_instance_norm_add_relu_lazy(x.data_ptr(), lazy.params.data_ptr(), int(lazy.relu), c.data_ptr(),
                                                   y.data_ptr(), ws.data_ptr(), n * ch, h * w, float(norm.eps),
                                                   _ffi.device_of(c), _ffi.stream_of(c))
        else:
            rc = L.dkt_instance_norm_add_relu(x.data_ptr(), c.data_ptr(), y.data_ptr(), ws.data_ptr(), n * ch, h * w,
                                              float(norm.eps), _ffi.device_of(c), _ffi.stream_of(c))
        _ffi.check(rc, "dkt_instance_norm_add_relu")
        return y
    if lazy is not None:
        x = lazy.materialize()
    return add_relu(x, norm_act(norm, c, True))


class LazyNorm:
    """[relu](instance_norm(raw)) that has not been evaluated: the raw convolution output and its per-plane
    (mean, 1/std).  Consumers fold the normalisation into their own pass (conv2d_fused(in_norm=...) reads it in its
    staging, dkt_instance_norm_add_relu_lazy applies it to the residual operand); ``materialize`` runs the plain pass."""

    def __init__(self, norm, raw, relu, stats=None):
        self.norm, self.raw, self.relu, self.stats = norm, raw, relu, stats
        self._params = None

    @property
    def params(self):
        """(mean, 1/std) per plane, computed on first use (a consumer that materialises never needs them)."""
        if self._params is None:
            self._params = instance_norm_params(self.norm, self.raw, self.stats)
        return self._params

    def materialize(self):
        return norm_act(self.norm, self.raw, self.relu)


def add_relu(a, b):
    """relu(a + b), one pass (dkt_add_relu)."""
    if _hip_ok(a) and _hip_ok(b) and a.shape == b.shape and a.is_contiguous() and b.is_contiguous():
        y = torch.empty_like(a)
        rc = _ffi.lib().dkt_add_relu(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(),
                                     _ffi.device_of(a), _ffi.stream_of(a))
        _ffi.check(rc, "dkt_add_relu")
        return y
    return F.relu(a + b)


class _Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state-dict keys) whose inference calls with stride 1 or 2
    go through dkt_stereo_amd.conv.conv2d (split-fp16 MFMA kernels; conv2d itself falls back to
    the vendor convolution for shapes it does not cover); autograd and CPU are plain torch."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and self.stride in ((1, 1), (2, 2)) and self.dilation == (1, 1)
                and self.groups == 1 and not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))):
            return conv2d(x, self)
        return super().forward(x)


def _make_norm(kind, channels, groups=None):
    if kind == 'group':
        return nn.GroupNorm(num_groups=groups if groups else channels // 8, num_channels=channels)
    if kind == 'batch':
        return nn.BatchNorm2d(channels)
    if kind == 'instance':
        return nn.InstanceNorm2d(channels)
    if kind == 'none':
        return nn.Sequential()
    raise ValueError("unknown norm_fn %r" % (kind,))


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super().__init__()
        self.conv1 = _Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = _Conv2d(planes, planes, kernel_size=3, padding=1)
        self.norm1 = _make_norm(norm_fn, planes)
        self.norm2 = _make_norm(norm_fn, planes)
        self.downsample = None
        if stride != 1 or in_planes != planes:
            # the projection's norm is registered under both names, as upstream does
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(_Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        lazy = x if isinstance(x, LazyNorm) else None
        # (trainable weights under autograd: the torch path, as _Conv2d.forward decides layer by layer)
        fuse = (FUSE_ENCODER and _hip_ok(lazy.raw if lazy is not None else x)
                and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())))
        if fuse and _plain_instance_norm(self.norm1) and _plain_instance_norm(self.norm2) and fused_eligible(self.conv2, True):
            # fnet: relu(norm1(.)) between the two layers lives in conv2's staging (no normalise pass, the
            # intermediate is read once for its statistics and once by conv2)
            # (round 4) the statistics each norm needs are accumulated by the convolution that produces its input
            es = EPILOGUE_STATS
            s1 = s2 = None
            if lazy is not None and lazy.relu and self.downsample is None and fused_eligible(self.conv1, True):
                if es and stats_eligible(self.conv1):                            # the block's input is normalised on the fly too
                    c1, s1 = conv2d_stats(lazy.raw, self.conv1, in_norm=lazy.params)
                else:
                    c1 = conv2d_fused(lazy.raw, self.conv1, in_norm=lazy.params)
            else:
                if lazy is not None:
                    x, lazy = lazy.materialize(), None
                if es and stats_eligible(self.conv1):
                    c1, s1 = conv2d_stats(x, self.conv1)
                else:
                    c1 = self.conv1(x)
            if self.downsample is not None:
                # the projection's norm (no ReLU) is applied inside the join
                if _plain_instance_norm(self.norm3):
                    if es and stats_eligible(self.downsample[0]):
                        raw, s3 = conv2d_stats(x, self.downsample[0])
                    else:
                        raw, s3 = self.downsample[0](x), None
                    x = LazyNorm(self.norm3, raw, relu=False, stats=s3)
                else:
                    x = conv_norm_act(self.downsample[0], self.norm3, x, False)
            p1 = instance_norm_params(self.norm1, c1, s1)
            if es and stats_eligible(self.conv2):
                c2, s2 = conv2d_stats(c1, self.conv2, in_norm=p1)
            else:
                c2 = conv2d_fused(c1, self.conv2, in_norm=p1)
            return norm_add_relu(self.norm2, x, c2, s2)
        if lazy is not None:
            x = lazy.materialize()
        y = conv_norm_act(self.conv1, self.norm1, x, True)
        if self.downsample is not None:
            x = conv_norm_act(self.downsample[0], self.norm3, x, False)
        if _plain_instance_norm(self.norm2):
            return norm_add_relu(self.norm2, x, self.conv2(y))      # norm2 + ReLU folded into the join
        if (fuse and _foldable(self.conv2, self.norm2, y) and fused_eligible(self.conv2) and x.shape[1] == self.conv2.weight.shape[0]):
            # cnet: norm2 folded into the weights, ReLU + residual join in the epilogue
            return conv2d_fused(y, _folded(self.conv2, self.norm2), relu=True, residual=x)
        y = conv_norm_act(self.conv2, self.norm2, y, True)
        return add_relu(x, y)


def _stage(in_planes, planes, norm_fn, stride):
    return nn.Sequential(ResidualBlock(in_planes, planes, norm_fn, stride=stride),
                         ResidualBlock(planes, planes, norm_fn, stride=1))


def _init_like_reference(module):
    # core/extractor.py:150-157
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
            if m.weight is not None:
                nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


#: the two output heads of a scale (hidden state, context; core/extractor.py:246-266) share their launches: both first
#: layers read the same tensor -- ONE convolution with the outputs concatenated --, the later layers run as pairs
#: (dkt_conv2d_f16s_pair).  Three launches per scale instead of six on a chain of small, latency-bound launches
PAIR_HEADS = True


def _merged_outputs(layers):
    """One layer computing the outputs of `layers` (same input, same filter geometry) side by side; cached on the first."""
    with _CACHE_LOCK:
        key = tuple(_tensor_key(t) for l in layers for t in (l.weight, l.bias))
        cache = layers[0].__dict__.setdefault("_dkt_merged", {})
        slot = str(layers[0].weight.device)
        hit = cache.get(slot)
        if hit is not None and hit.key == key:
            return hit
        m = _Folded()
        with torch.no_grad():
            m.weight = torch.cat([l.weight.detach().float() for l in layers], 0).contiguous()
            m.bias = torch.cat([(l.bias.detach().float() if l.bias is not None else
                                 torch.zeros(l.weight.shape[0], device=l.weight.device)) for l in layers], 0).contiguous()
        m.padding, m.stride, m.dilation, m.groups = layers[0].padding, layers[0].stride, layers[0].dilation, layers[0].groups
        m.key = key
        cache[slot] = m
        return m


def _same_geometry(a, b):
    return (tuple(a.weight.shape[1:]) == tuple(b.weight.shape[1:]) and tuple(a.padding) == tuple(b.padding)
            and tuple(a.stride) == tuple(b.stride) == (1, 1) and tuple(a.dilation) == tuple(b.dilation) == (1, 1)
            and a.groups == b.groups == 1)


def paired_heads(heads, x):
    """[head(x) for head in heads] for the context encoder's two heads of one scale, in shared launches; None when the pair
    does not have the expected form (then the caller runs them one after the other)."""
    if not (PAIR_HEADS and FUSE_ENCODER and len(heads) == 2 and _hip_ok(x)):
        return None
    if all(isinstance(h, nn.Conv2d) for h in heads):                       # the 1/16 scale: one 3x3 layer per head
        a, b = heads
        if not (_same_geometry(a, b) and a.padding_mode == b.padding_mode == 'zeros'
                and not (torch.is_grad_enabled() and (a.weight.requires_grad or b.weight.requires_grad))):
            return None
        y = conv2d(x, _merged_outputs([a, b]))
        return list(y.split([a.weight.shape[0], b.weight.shape[0]], dim=1))
    if not all(isinstance(h, nn.Sequential) and len(h) == 2 and isinstance(h[0], ResidualBlock) and isinstance(h[1], nn.Conv2d)
               for h in heads):
        return None
    (ra, fa), (rb, fb) = heads
    if ra.downsample is not None or rb.downsample is not None or any(p.requires_grad and torch.is_grad_enabled()
                                                                     for h in heads for p in h.parameters()):
        return None
    if not all(_foldable(r.conv1, r.norm1, x) and _foldable(r.conv2, r.norm2, x) for r in (ra, rb)):
        return None
    a1, b1 = _folded(ra.conv1, ra.norm1), _folded(rb.conv1, rb.norm1)
    a2, b2 = _folded(ra.conv2, ra.norm2), _folded(rb.conv2, rb.norm2)
    if not (_same_geometry(a1, b1) and pair_eligible(a2, b2) and fused_eligible(a2) and fused_eligible(b2) and pair_eligible(fa, fb)
            and a2.weight.shape[0] == x.shape[1] == b2.weight.shape[0] and fa.padding_mode == fb.padding_mode == 'zeros'):
        return None
    na = a1.weight.shape[0]
    y = conv2d(x, _merged_outputs([a1, b1]), relu=True)                    # relu(norm1(conv1(x))) of both blocks
    ya, yb = conv2d_fused_pair((y[:, :na], a2, True, x), (y[:, na:], b2, True, x))      # relu(x + relu(norm2(conv2(.))))
    return conv2d_pair((ya, fa, False), (yb, fb, False))


class _Trunk(nn.Module):
    """conv1/norm1 + layer1..3, shared by both encoders (1/2^downsample resolution)."""

    def _build_trunk(self, norm_fn, downsample):
        self.norm_fn = norm_fn
        self.downsample = downsample
        self.norm1 = _make_norm(norm_fn, 64, groups=8)
        self.conv1 = _Conv2d(3, 64, kernel_size=7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, norm_fn, 1)
        self.layer2 = _stage(64, 96, norm_fn, 1 + (downsample > 1))
        self.layer3 = _stage(96, 128, norm_fn, 1 + (downsample > 0))

    def _layer1_c8_kind(self, x):
        """'batch' / 'instance' when conv1 + layer1 can run on the C8S convolution (conv_c8.hip): stride-1 stem, two plain
        64 -> 64 residual blocks, one norm kind throughout (folded eval-mode BatchNorm, or affine-free instance norm)."""
        from . import conv as _conv
        if not (C8_ENCODER and FUSE_ENCODER and torch.is_tensor(x) and x.dim() == 4 and _hip_ok(x)):
            return None
        if x.shape[1] > 4 or x.shape[2] * x.shape[3] < C8_ENCODER_MIN_PIXELS or _conv.get_backend() != "f16x3":
            return None
        convs = [self.conv1] + [c for b in self.layer1 for c in (b.conv1, b.conv2)]
        norms = [self.norm1] + [n for b in self.layer1 for n in (b.norm1, b.norm2)]
        if (self.conv1.stride != (1, 1) or tuple(self.conv1.weight.shape[2:]) != (7, 7) or self.conv1.padding != (3, 3)
                or self.conv1.weight.shape[0] != 64 or any(b.downsample is not None for b in self.layer1)):
            return None
        if any(getattr(c, "dkt_in_exp", 0) or c.padding_mode != 'zeros' or (torch.is_grad_enabled() and c.weight.requires_grad)
               for c in convs):
            return None
        if any(tuple(c.weight.shape) != (64, 64, 3, 3) or c.stride != (1, 1) or c.dilation != (1, 1) or c.groups != 1
               for c in convs[1:]):
            return None
        if all(_plain_instance_norm(n) for n in norms):
            return 'instance'
        if all(isinstance(n, nn.BatchNorm2d) and not n.training and n.track_running_stats for n in norms):
            return 'batch'
        return None

    def _c8_buffers(self, x):
        """Two C8S tensors of the full-resolution stage, kept between calls (their zero border is written once)."""
        from . import conv_c8 as c8
        B, _, H, W = x.shape
        if torch.cuda.is_current_stream_capturing():
            return c8.ActC8(B, 64, H, W, x.device), c8.ActC8(B, 64, H, W, x.device)
        key = (B, H, W, str(x.device))
        with _CACHE_LOCK:
            cache = self.__dict__.setdefault("_dkt_c8_buf", {})
            hit = cache.get(str(x.device))
            if hit is None or hit[0] != key:
                hit = cache[str(x.device)] = (key, c8.ActC8(B, 64, H, W, x.device), c8.ActC8(B, 64, H, W, x.device))
        return hit[1], hit[2]

    def _layer1_c8(self, x, kind):
        """conv1 / norm1 / relu + layer1 (core/extractor.py:140-146, :167-173 with ResidualBlock :52-60) on conv_c8: every
        3x3 layer reads the pre-split C8S operand its producer's epilogue (or the one-pass instance-norm glue) wrote."""
        from . import conv_c8 as c8
        x = x if x.is_contiguous() else x.contiguous()
        A, Bf = self._c8_buffers(x)
        B, _, H, W = x.shape
        new = lambda: torch.empty((B, 64, H, W), device=x.device, dtype=torch.float32)
        cfg = C8_ENCODER_CFG
        if kind == 'batch':
            cur = c8.stem7_dual(x, _folded(self.conv1, self.norm1), new(), A, relu=True)
            for i, blk in enumerate(self.layer1):
                last = i + 1 == len(self.layer1)
                c8.conv2d_c8([A], _folded(blk.conv1, blk.norm1), relu=True, out_c8=Bf, cfg=cfg)
                # the join reads `cur` and writes it in place (each element by the thread that read it)
                c8.residual_c8([Bf], _folded(blk.conv2, blk.norm2), cur, relu=True, out=cur, out_c8=None if last else A, cfg=cfg)
            return cur
        raw0 = conv2d(x, self.conv1)
        res, res_p = raw0, instance_norm_params(self.norm1, raw0)       # residual operand: relu(norm1(raw0)), evaluated in the join
        c8.norm_join_c8(raw0, res_p, True, dst=A)
        for i, blk in enumerate(self.layer1):
            last = i + 1 == len(self.layer1)
            c1 = c8.conv2d_c8([A], blk.conv1, out=new(), cfg=cfg)
            c8.norm_join_c8(c1, instance_norm_params(blk.norm1, c1), True, dst=Bf)
            c2 = c8.conv2d_c8([Bf], blk.conv2, out=c1, cfg=cfg)      # (c1 has been consumed)
            y = new()
            c8.norm_join_c8(c2, instance_norm_params(blk.norm2, c2), True, a=res, a_params=res_p, a_relu=True,
                            y=y, dst=None if last else A)
            res, res_p = y, None
        return res

    def _trunk_begin(self, x):
        """conv1 / norm1 / relu + layer1 (core/extractor.py:167-173): the full-resolution stage."""
        kind = self._layer1_c8_kind(x)
        if kind is not None:
            return self._layer1_c8(x, kind)
        if (FUSE_ENCODER and _plain_instance_norm(self.norm1) and _hip_ok(x)
                and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))):
            # fnet: the stem's normalise + ReLU pass is folded into its two consumers (layer1.0.conv1's staging, the
            # residual operand of layer1.0's join)
            x = LazyNorm(self.norm1, self.conv1(x), relu=True)
        else:
            x = conv_norm_act(self.conv1, self.norm1, x, True)
        return self.layer1(x)

    def _trunk(self, x, begun=None):
        """`begun`: the result of _trunk_begin(x) when the caller has enqueued that stage already (RAFTStereo._encode puts the
        context encoder's full-resolution stage on the device BEFORE it enqueues the feature encoder on the second stream)."""
        return self.layer3(self.layer2(self._trunk_begin(x) if begun is None else begun))


class BasicEncoder(_Trunk):
    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0, downsample=3):
        super().__init__()
        self._build_trunk(norm_fn, downsample)
        self.conv2 = _Conv2d(128, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init_like_reference(self)

    def forward(self, x, dual_inp=False):
        pair = isinstance(x, (tuple, list))
        if pair:
            batch_dim = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = self.conv2(self._trunk(x))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        if pair:
            x = x.split(split_size=batch_dim, dim=0)
        return x


def _tensors_of(obj):
    if torch.is_tensor(obj):
        return [obj]
    if isinstance(obj, (list, tuple)):
        return [t for o in obj for t in _tensors_of(o)]
    return []


class MultiBasicEncoder(_Trunk):
    def __init__(self, output_dim=[128], norm_fn='batch', dropout=0.0, downsample=3):
        super().__init__()
        self._build_trunk(norm_fn, downsample)
        self.layer4 = _stage(128, 128, norm_fn, 2)
        self.layer5 = _stage(128, 128, norm_fn, 2)
        self.outputs08 = nn.ModuleList([
            nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), _Conv2d(128, dim[2], 3, padding=1))
            for dim in output_dim])
        self.outputs16 = nn.ModuleList([
            nn.Sequential(ResidualBlock(128, 128, norm_fn, stride=1), _Conv2d(128, dim[1], 3, padding=1))
            for dim in output_dim])
        self.outputs32 = nn.ModuleList([_Conv2d(128, dim[0], 3, padding=1) for dim in output_dim])
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init_like_reference(self)

    @staticmethod
    def _heads(heads, x):
        """[f(x) for f in heads] (core/extractor.py:284-296), the two heads of a scale in shared launches where possible."""
        outs = paired_heads(list(heads), x)
        return outs if outs is not None else [f(x) for f in heads]

    def _heads_on_streams(self, x, num_layers, post):
        """The output heads of the three scales beside the trunk's coarse stages: the 1/8 and 1/16 layers are a few
        dozen tiles each -- 30 us per layer whatever their size -- and left ~0.6 ms of the pair with one such kernel at
        a time on the device.  main: layer4 -> layer5 -> 1/16 heads; stream a: 1/4 heads; stream b: 1/8 heads."""
        from .update import _side_stream
        main = torch.cuda.current_stream(x.device)
        sa, sb = _side_stream(x.device, slot=2), _side_stream(x.device, slot=3)
        sa.wait_stream(main)
        with torch.cuda.stream(sa):
            s08 = post(0, self._heads(self.outputs08, x))
        y = self.layer4(x)
        sb.wait_stream(main)
        with torch.cuda.stream(sb):
            s16 = post(1, self._heads(self.outputs16, y))
        scales = [s08, s16]
        if num_layers >= 3:
            z = self.layer5(y)
            scales.append(post(2, self._heads(self.outputs32, z)))
        main.wait_stream(sa)
        main.wait_stream(sb)
        for t in _tensors_of(s08) + _tensors_of(s16):    # allocated on the side streams, consumed on this one from here on
            t.record_stream(main)
        return scales

    def forward(self, x, dual_inp=False, num_layers=3, head_post=None, begun=None):
        """head_post(i, [hidden_i, context_i]) -> anything: an optional per-scale epilogue of the caller (RAFT-Stereo's tanh /
        relu / context_zqr convolution, raft_stereo.py:103-106) that then runs on the stream of that scale's heads.
        begun: _trunk_begin(x) of this very input, already enqueued by the caller."""
        x = self._trunk(x, begun)
        v = None
        if dual_inp:
            v = x
            x = x[:(x.shape[0] // 2)]
        post = head_post if head_post is not None else (lambda i, outs: outs)
        if CNET_STREAMS and num_layers >= 2 and _hip_ok(x) and not torch.cuda.is_current_stream_capturing():
            scales = self._heads_on_streams(x, num_layers, post)
        else:
            scales = [post(0, self._heads(self.outputs08, x))]
            if num_layers >= 2:
                y = self.layer4(x)
                scales.append(post(1, self._heads(self.outputs16, y)))
            if num_layers >= 3:
                z = self.layer5(y)
                scales.append(post(2, self._heads(self.outputs32, z)))
        if dual_inp:
            scales.append(v)
        return tuple(scales)
