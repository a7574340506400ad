"""Single entry point for every 2-D convolution on the hot path.

``conv2d(x, layer, relu=False)`` takes anything with ``weight``/``bias``/
``padding`` attributes (an ``nn.Conv2d`` or the merged z|r pair built by
``ConvGRU``).  Backend ``"miopen"`` is the vendor convolution reached through
``torch.nn.functional.conv2d`` (fp32, the parity path).  The update block's
convolutions are ~98 % of an iteration; keeping them behind
this one function is what lets later rounds swap in hand-written MFMA
implicit-GEMM kernels without touching the operator code.
"""
import torch
import torch.nn.functional as F

_BACKEND = "miopen"


def set_backend(name):
    global _BACKEND
    if name not in ("miopen",):
        raise ValueError("unknown conv backend %r" % (name,))
    _BACKEND = name


def get_backend():
    return _BACKEND


def conv2d(x, layer, relu=False):
    pad = layer.padding
    y = F.conv2d(x, layer.weight, layer.bias, stride=1, padding=pad)
    if relu:
        y = torch.relu_(y)
    return y
