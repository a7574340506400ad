"""Helpers of the reference's ``core/utils/utils.py`` that the harness needs."""
import torch
import torch.nn.functional as F


def coords_grid(batch, ht, wd):
    """Pixel coordinate grid (B,2,H,W), channel 0 = x, channel 1 = y
    (core/utils/utils.py:77-80)."""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing='ij')
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


class InputPadder:
    """Replicate-pads a (…,H,W) image so that H and W become multiples of ``divis_by``
    and crops results back (behaviour of core/utils/utils.py:7-26; the evaluator uses
    ``divis_by=32``, tools/evaluate_stereo.py:124).  ``mode='sintel'`` splits the padding
    evenly between both sides; any other mode pads width evenly and height at the bottom."""

    def __init__(self, dims, mode='sintel', divis_by=8):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        extra_h = -self.ht % divis_by
        extra_w = -self.wd % divis_by
        left, right = extra_w // 2, extra_w - extra_w // 2
        if mode == 'sintel':
            top, bottom = extra_h // 2, extra_h - extra_h // 2
        else:
            top, bottom = 0, extra_h
        self._pad = [left, right, top, bottom]      # F.pad order for the last two dims

    def pad(self, *inputs):
        for x in inputs:
            assert x.ndim == 4
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def unpad(self, x):
        assert x.ndim == 4
        left, right, top, bottom = self._pad
        h, w = x.shape[-2:]
        return x[..., top:h - bottom, left:w - right]
