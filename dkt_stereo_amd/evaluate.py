"""End-to-end evaluation harness: files on disk -> padded pair -> model -> un-padded disparity -> EPE / D1.

The counterpart of ``validate_kitti`` (tools/evaluate_stereo.py:108-170) and of the sample convention of
``core/stereo_datasets.py`` (:73-137: images as float CHW in [0, 255], ground truth as ``flow = -disparity``
in one channel, a validity mask), reduced to what inference needs.  Data sets and checkpoints are not part of
this library; what is here is the chain that ties ``frame_utils`` (readers), ``utils.InputPadder``, a model of
this package (``RAFTStereo``, ``GWCNet``; anything with the reference's ``forward(image1, image2, iters,
test_mode=True) -> (_, flow_up)``) and the metric together, so that it is exercised and pinned.

    pairs = kitti_pairs(root)                       # [(left.png, right.png, disp_occ.png), ...]
    res = validate(model, pairs, iters=32)          # {'epe': ..., 'd1': ..., 'fps': ...}
"""
import glob
import os
import time

import numpy as np
import torch

from . import frame_utils
from .utils import InputPadder


def load_sample(left, right, disp, reader=None):
    """One evaluation sample as core/stereo_datasets.py:73-137 builds it (no augmentation):
    image1, image2 float (3,H,W) in [0,255]; flow_gt = -disparity (1,H,W); valid (H,W) float."""
    img1 = np.array(frame_utils.read_gen(left)).astype(np.uint8)
    img2 = np.array(frame_utils.read_gen(right)).astype(np.uint8)
    if img1.ndim == 2:                                   # grayscale: replicate (stereo_datasets.py:91-93)
        img1 = np.tile(img1[..., None], (1, 1, 3))
        img2 = np.tile(img2[..., None], (1, 1, 3))
    img1, img2 = img1[..., :3], img2[..., :3]
    d = (reader or frame_utils.readDispKITTI)(disp)
    if isinstance(d, tuple):
        d, valid = d
    else:
        valid = (d < 512) & (d > 0)                       # stereo_datasets.py:71
    d = np.asarray(d, np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float()      # noqa: E731
    return t(img1), t(img2), -torch.from_numpy(d)[None], torch.from_numpy(np.asarray(valid)).float()


def kitti_pairs(root, split="2015"):
    """File triples of the KITTI training layout (stereo_datasets.py:281-306)."""
    sub = {"2015": ("KITTI_2015", "image_2", "image_3", "disp_occ_0"),
           "2012": ("KITTI_2012", "colored_0", "colored_1", "disp_occ")}[split]
    base = os.path.join(root, sub[0], "training")
    lefts = sorted(glob.glob(os.path.join(base, sub[1], "*_10.png")))
    rights = sorted(glob.glob(os.path.join(base, sub[2], "*_10.png")))
    disps = sorted(glob.glob(os.path.join(base, sub[3], "*_10.png")))
    return list(zip(lefts, rights, disps))


def pair_metrics(flow_pr, flow_gt, valid_gt, maxdisp=192):
    """EPE and the >3 px outlier mask over the valid pixels of one image (evaluate_stereo.py:149-157).
    flow_pr, flow_gt: (1,H,W) (negative disparities); valid_gt: (H,W).  Returns (epe, outlier vector)."""
    assert flow_pr.shape == flow_gt.shape, (flow_pr.shape, flow_gt.shape)
    epe = torch.sum((flow_pr - flow_gt) ** 2, dim=0).sqrt().flatten()
    val = (valid_gt.reshape(-1) >= 0.5) & (flow_gt[0].reshape(-1) > -maxdisp) & (flow_gt[0].reshape(-1) < 0)
    return epe[val].mean().item(), (epe > 3.0)[val].cpu().numpy()


@torch.no_grad()
def validate(model, samples, iters=32, maxdisp=192, divide_factor=32, device="cuda", reader=None, keep=False):
    """validate_kitti's loop over `samples` (file triples, or already loaded 4-tuples): pad to a multiple of
    `divide_factor` (replicate), forward in test_mode, un-pad, EPE / D1 against the ground truth.
    Returns {'epe', 'd1' (percent), 'fps', 'n'} (+ 'predictions' with keep=True)."""
    model.eval()
    out_list, epe_list, elapsed, preds = [], [], [], []
    for s in samples:
        image1, image2, flow_gt, valid_gt = s if torch.is_tensor(s[0]) else load_sample(*s, reader=reader)
        image1 = image1[None].to(device)
        image2 = image2[None].to(device)
        padder = InputPadder(image1.shape, divis_by=divide_factor)
        image1, image2 = padder.pad(image1, image2)
        if image1.is_cuda:
            torch.cuda.synchronize()
        t0 = time.time()
        _, flow_pr = model(image1, image2, iters=iters, test_mode=True)
        if flow_pr.is_cuda:
            torch.cuda.synchronize()
        elapsed.append(time.time() - t0)
        flow_pr = padder.unpad(flow_pr).cpu().squeeze(0)
        epe, out = pair_metrics(flow_pr, flow_gt, valid_gt, maxdisp)
        epe_list.append(epe)
        out_list.append(out)
        if keep:
            preds.append(flow_pr)
    res = {"epe": float(np.mean(epe_list)), "d1": 100.0 * float(np.mean(np.concatenate(out_list))),
           "fps": 1.0 / float(np.mean(elapsed[1:] or elapsed)), "n": len(epe_list)}
    if keep:
        res["predictions"] = preds
    return res
