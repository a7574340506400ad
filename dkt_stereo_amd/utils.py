"""Small helpers of the reference's ``core/utils/utils.py`` that the harness needs."""
import torch
import torch.nn.functional as F


def coords_grid(batch, ht, wd):
    """core/utils/utils.py:77-80: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing='ij')
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


class InputPadder:
    """core/utils/utils.py:7-26: replicate-pad so H, W are divisible by `divis_by`
    (tools/evaluate_stereo.py:124 uses divis_by=32)."""

    def __init__(self, dims, mode='sintel', divis_by=8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // divis_by) + 1) * divis_by - self.ht) % divis_by
        pad_wd = (((self.wd // divis_by) + 1) * divis_by - self.wd) % divis_by
        if mode == 'sintel':
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        assert all((x.ndim == 4) for x in inputs)
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def unpad(self, x):
        assert x.ndim == 4
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]
