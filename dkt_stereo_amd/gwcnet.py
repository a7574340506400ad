"""GwcNet single-shot inference harness around this library's cost-volume kernels (SURVEY.md 8a-13,
BASELINE.json configs[4]).

Counterpart of ``meta_arch/gwcnet/gwc_main.py`` (``GWCNet.forward`` :279-326, ``cost_regularization``
:234-277) and ``meta_arch/gwcnet/submodules.py`` with identical sub-module / parameter names
(``feature_extraction.firstconv.0.0.weight``, ``dres2.conv5.1.running_mean``, ``classif3.2.weight`` ...),
so the reference's checkpoints load with ``strict=True``.  Like the reference's own ``GWCNet`` it can be
pointed at ``dkt_stereo_amd.submodule`` for its volumes; this class exists so that the whole path runs, is
timed and is parity-checked where the reference is not present.

What runs where (inference, ``test_mode=True``):
  * 2-D feature extraction: every 3x3 / 1x1 convolution with its eval-mode BatchNorm folded in runs on
    dkt_conv2d_f16s[_strided]; the dilated layer4 stays on the vendor library;
  * group-wise correlation (40 groups) + concatenation volume: ONE 64-channel buffer written by
    dkt_gwc_volume / dkt_concat_volume (the reference's torch.cat of a 250 MB and a 150 MB tensor,
    gwc_main.py:315, never happens);
  * 3-D aggregation (dres0-4, classif3): dense Conv3d / ConvTranspose3d on the vendor library (out of
    this library's scope), eval-mode BatchNorm3d folded into the weights;
  * soft-argmin: trilinear x4 up-sampling, softmax over the 192 disparities, expectation
    (gwcnet/submodules.py:18-22).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .extractor import conv_norm_act
from .submodule import build_gwc_concat_volume, build_gwc_volume


def convbn(cin, cout, k, stride, pad, dilation):
    """gwcnet/submodules.py:6-9."""
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=dilation if dilation > 1 else pad,
                                   dilation=dilation, bias=False), nn.BatchNorm2d(cout))


def convbn_3d(cin, cout, k, stride, pad):
    """gwcnet/submodules.py:12-15."""
    return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False), nn.BatchNorm3d(cout))


def disparity_regression(x, maxdisp):
    """gwcnet/submodules.py:18-22: expectation over the disparity axis, (B,D,H,W) -> (B,H,W)."""
    assert x.dim() == 4
    d = torch.arange(0, maxdisp, dtype=x.dtype, device=x.device).view(1, maxdisp, 1, 1)
    return torch.sum(x * d, 1, keepdim=False)


def _cbn(seq, x, relu):
    """convbn [+ ReLU] with the BatchNorm folded into the convolution where this library's kernel applies."""
    return conv_norm_act(seq[0], seq[1], x, relu)


class BasicBlock(nn.Module):
    """gwcnet/submodules.py:62-84."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = _cbn(self.conv2, _cbn(self.conv1[0], x, True), False)
        if self.downsample is not None:
            x = _cbn(self.downsample, x, False)
        return out + x


class feature_extraction(nn.Module):
    """gwc_main.py:60-118: ResNet-like 2-D features at 1/4 resolution, 320 channels for the group-wise
    correlation (+ 12 for the concatenation volume)."""

    def __init__(self, concat_feature=False, concat_feature_channel=12):
        super().__init__()
        self.concat_feature = concat_feature
        self.inplanes = 32
        self.firstconv = nn.Sequential(convbn(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                       convbn(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._make_layer(32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(128, 3, 1, 1, 2)
        if concat_feature:
            self.lastconv = nn.Sequential(convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, concat_feature_channel, kernel_size=1, padding=0, stride=1, bias=False))

    def _make_layer(self, planes, blocks, stride, pad, dilation):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down, pad, dilation)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        for i in (0, 2, 4):
            x = _cbn(self.firstconv[i], x, True)
        x = self.layer1(x)
        l2 = self.layer2(x)
        l3 = self.layer3(l2)
        l4 = self.layer4(l3)
        feat = torch.cat((l2, l3, l4), dim=1)
        if not self.concat_feature:
            return {"gwc_feature": feat}
        from .conv import conv2d
        cat = _cbn(self.lastconv[0], feat, True)
        cat = conv2d(cat, self.lastconv[2]) if cat.is_cuda and not torch.is_grad_enabled() else self.lastconv[2](cat)
        return {"gwc_feature": feat, "concat_feature": cat}


def _fold3d(conv, bn):
    """(weight, bias) of bn(conv(x)) for an eval-mode BatchNorm3d: w * g per output channel, b = beta - mean * g.
    ConvTranspose3d keeps its output channels on axis 1."""
    g = (bn.weight.double() * torch.rsqrt(bn.running_var.double() + bn.eps))
    shape = (1, -1, 1, 1, 1) if isinstance(conv, nn.ConvTranspose3d) else (-1, 1, 1, 1, 1)
    w = (conv.weight.double() * g.view(shape)).float()
    b = (bn.bias.double() - bn.running_mean.double() * g).float()
    return w, b


def _cbn3(seq, x, relu=False):
    conv, bn = seq[0], seq[1]
    if bn.training:
        y = bn(conv(x))
    else:
        w, b = _fold3d(conv, bn)
        if isinstance(conv, nn.ConvTranspose3d):
            y = F.conv_transpose3d(x, w, b, stride=conv.stride, padding=conv.padding, output_padding=conv.output_padding)
        else:
            y = F.conv3d(x, w, b, stride=conv.stride, padding=conv.padding)
    return F.relu_(y) if relu else y


class hourglass(nn.Module):
    """gwc_main.py:121-152."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(c * 4, c * 2, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(c * 2, c, 3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(c))
        self.redir1 = convbn_3d(c, c, 1, 1, 0)
        self.redir2 = convbn_3d(c * 2, c * 2, 1, 1, 0)

    def forward(self, x):
        c1 = _cbn3(self.conv1[0], x, True)
        c2 = _cbn3(self.conv2[0], c1, True)
        c3 = _cbn3(self.conv3[0], c2, True)
        c4 = _cbn3(self.conv4[0], c3, True)
        c5 = F.relu_(_cbn3(self.conv5, c4) + _cbn3(self.redir2, c2))
        return F.relu_(_cbn3(self.conv6, c5) + _cbn3(self.redir1, x))


#: configs of the reference for this model (maxdisp 192, both volumes)
def make_args(**over):
    cfg = dict(model="GWCNet", maxdisp=192, use_concat_volume=True, mixed_precision=False)
    cfg.update(over)
    return SimpleNamespace(**cfg)


class GWCNet(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args = args if args is not None else make_args()
        self.maxdisp = args.maxdisp
        self.use_concat_volume = args.use_concat_volume
        self.num_groups = 40
        self.concat_channels = 12 if self.use_concat_volume else 0
        self.feature_extraction = feature_extraction(concat_feature=self.use_concat_volume,
                                                     concat_feature_channel=12)
        cin = self.num_groups + 2 * self.concat_channels
        self.dres0 = nn.Sequential(convbn_3d(cin, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        for name in ("classif0", "classif1", "classif2", "classif3"):
            setattr(self, name, nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                              nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))

    def freeze_bn(self):
        pass

    def cost_regularization(self, cost):
        """gwc_main.py:234-277, eval branch: only the last classifier contributes to the result."""
        c0 = _cbn3(self.dres0[2], _cbn3(self.dres0[0], cost, True), True)
        c0 = _cbn3(self.dres1[2], _cbn3(self.dres1[0], c0, True)) + c0
        out3 = self.dres4(self.dres3(self.dres2(c0)))
        cost3 = self.classif3[2](_cbn3(self.classif3[0], out3, True))
        cost3 = F.interpolate(cost3, scale_factor=4, mode='trilinear', align_corners=False)
        pred3 = F.softmax(torch.squeeze(cost3, 1), dim=1)
        return -disparity_regression(pred3, self.maxdisp).unsqueeze(1)

    def build_volume(self, featL, featR):
        """gwc_main.py:310-317."""
        d = self.maxdisp // 4
        if self.use_concat_volume:
            return build_gwc_concat_volume(featL["gwc_feature"], featR["gwc_feature"],
                                           featL["concat_feature"], featR["concat_feature"], d, self.num_groups)
        return build_gwc_volume(featL["gwc_feature"], featR["gwc_feature"], d, self.num_groups)

    @torch.no_grad()
    def forward(self, imgL, imgR, iters=None, flow_init=None, test_mode=False):
        if not test_mode or self.training:
            raise NotImplementedError("dkt_stereo_amd.GWCNet is the inference path (eval(), test_mode=True)")
        imgL = (2 * (imgL / 255.0) - 1.0).contiguous()
        imgR = (2 * (imgR / 255.0) - 1.0).contiguous()
        # both images through the shared-weight extractor as one batch of two
        B = imgL.shape[0]
        feats = self.feature_extraction(torch.cat([imgL, imgR], 0))
        featL = {k: v[:B] for k, v in feats.items()}
        featR = {k: v[B:] for k, v in feats.items()}
        return None, self.cost_regularization(self.build_volume(featL, featR))
