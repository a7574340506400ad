"""CPU tests (no GPU): oracle vs the slow-fast fixtures, host-side logic added in round 2
(thread-local harness switches and backend override, per-device weight caches, InputPadder)."""
import json
import os
import threading

import numpy as np
import pytest
import torch

import _cases
import _synth
from oracle import torch_oracle as to

T = torch.from_numpy


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.mark.parametrize("name", ["small", "sf3", "sf2"])
def test_igev_loop_oracle_all_schedules(name, golden):
    """igev_stereo.py:199-210 incl. the slow-fast schedule (:204-207), 3 and 2 GRU layers."""
    from test_layout import update_block_shapes
    c = _cases.IGEV_LOOP_CASES[name]
    cfg = _cases.igev_loop_cfg(c)
    shapes = update_block_shapes(True, cfg)
    sd = _synth.torch_state_dict({"update_block." + k: v for k, v in shapes.items()}, c["seed"])
    m1, m2, geo, disp0, coords, net, inp = _cases.igev_loop_inputs(c)
    tinp = [list(T(x).split(128, dim=1)) for x in inp]
    d, m = to.igev_iterations(sd, cfg, T(m1), T(m2), T(geo), T(disp0), [T(x) for x in net], tinp, c["iters"])
    g = golden("igev_loop")
    assert maxabs(d.numpy(), g[name + "/disp"]) <= 1e-4
    assert maxabs(m.numpy(), g[name + "/mask"]) <= 1e-4


def test_slow_fast_changes_the_result(golden):
    """Guards the fixtures themselves: the slow-fast schedule is not a no-op."""
    g = golden("igev_loop")
    assert g["sf3/disp"].shape == g["small/disp"].shape
    a, b = _cases.IGEV_LOOP_CASES["sf3"], _cases.IGEV_LOOP_CASES["small"]
    assert a["slow_fast"] and not b["slow_fast"]


@pytest.mark.parametrize("name", list(_cases.E2E_SLOWFAST_CASES))
def test_raft_slow_fast_oracle(name, golden):
    """raft_stereo.py:156-159 with 3 and 2 GRU layers."""
    c = _cases.E2E_SLOWFAST_CASES[name]
    from dkt_stereo_amd.raft_stereo import BASE_CONFIG, RAFTStereo, make_args
    over = dict(slow_fast_gru=True, n_gru_layers=c["n"])
    shapes = _synth.shapes_of(RAFTStereo(make_args(**over)))
    sd = _synth.torch_state_dict(shapes, _cases.E2E_WEIGHT_SEED)
    i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
    lo, up = to.raft_stereo_forward(sd, {**BASE_CONFIG, **over}, T(i1), T(i2), c["iters"])
    g = golden("raft_e2e")
    assert maxabs(up.numpy(), g[name + "/flow_up"]) <= 1e-3
    assert maxabs(lo.numpy()[:, :1], g[name + "/flow_lo"]) <= 1e-3


def test_igev_kitti_fixture_is_pinned():
    """The 184x312 / 32-iteration loop fixture is too slow to re-run on CPU in this suite; its pin
    (oracle == reference, 0.0) was recorded when it was generated."""
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "MANIFEST.json")))["pins"]
    for k in ("torch.igev_iterations[kitti].disp", "torch.igev_iterations[kitti].mask"):
        assert pins[k]["max_abs"] <= pins[k]["bound"]


def test_harness_switches_are_thread_local():
    from dkt_stereo_amd.update import _HARNESS, harness
    seen = {}
    gate = threading.Barrier(2)

    def worker(name, val):
        with harness(inplace_state=val, side_stream=False):
            gate.wait()
            seen[name] = (_HARNESS.inplace_state, _HARNESS.side_stream)
            gate.wait()
        seen[name + "_after"] = (_HARNESS.inplace_state, _HARNESS.side_stream)

    ts = [threading.Thread(target=worker, args=("a", True)), threading.Thread(target=worker, args=("b", False))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen["a"] == (True, False) and seen["b"] == (False, False)
    assert seen["a_after"] == (False, None) and seen["b_after"] == (False, None)
    assert (_HARNESS.inplace_state, _HARNESS.side_stream) == (False, None)


def test_backend_override_is_thread_local():
    from dkt_stereo_amd import conv
    base = conv.get_backend()
    got = {}

    def worker():
        got["before"] = conv.get_backend()
        with conv.use_backend("miopen"):
            got["inside"] = conv.get_backend()
        got["after"] = conv.get_backend()

    with conv.use_backend("f16x2"):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert conv.get_backend() == "f16x2"
    assert conv.get_backend() == base
    assert got == {"before": base, "inside": "miopen", "after": base}
    with pytest.raises(ValueError):
        conv.set_backend("fp8")


def test_merged_zr_cache_is_per_device_and_versioned():
    """ConvGRU's merged z|r weights: one entry per device, rebuilt when a parameter is written."""
    from dkt_stereo_amd.update import ConvGRU
    gru = ConvGRU(8, 8)
    a = gru._merged_zr()
    assert a is gru._merged_zr()
    assert torch.equal(a.weight, torch.cat([gru.convz.weight, gru.convr.weight], 0))
    with torch.no_grad():
        gru.convr.bias.add_(1.0)
    b = gru._merged_zr()
    assert b is not a and torch.equal(b.bias[8:], gru.convr.bias)
    assert list(gru._zr_cache) == ["cpu"]
    # the calibrated activation exponent lives on convz: it survives the rebuild of the merged layer
    b.dkt_in_exp = -3
    assert gru.convz.dkt_in_exp == -3
    with torch.no_grad():
        gru.convz.weight.mul_(2.0)
    assert gru._merged_zr() is not b and gru._merged_zr().dkt_in_exp == -3


def test_few_output_kernel_is_bounded_by_what_it_can_stage():
    """few_eligible (the 256 -> 2 / 256 -> 1 head layers) stops where conv_direct.hip's launch_few would exceed 160 KB of LDS;
    wider layers stay on the general kernel instead of raising DKT_E_UNSUPPORTED."""
    from dkt_stereo_amd import conv
    mk = lambda cin, cout: torch.nn.Conv2d(cin, cout, 3, padding=1)
    assert conv.few_eligible(mk(256, 2)) and conv.few_eligible(mk(256, 1)) and conv.few_eligible(mk(256, 4))
    assert conv.few_eligible(mk(552, 2)) and not conv.few_eligible(mk(553, 2))
    assert conv.few_eligible(mk(272, 3)) and not conv.few_eligible(mk(280, 4))
    assert conv.few_eligible(mk(1104, 1)) and not conv.few_eligible(mk(1105, 1))
    assert not conv.few_eligible(mk(64, 5))


@pytest.mark.parametrize("dims,div,mode", [((375, 1242), 32, "kitti"), ((540, 960), 32, "sintel"), ((736, 1248), 32, "kitti"),
                                           ((5, 7), 8, "sintel"), ((64, 64), 8, "other"), ((1, 1), 32, "sintel")])
def test_input_padder(dims, div, mode):
    """core/utils/utils.py:7-26: replicate padding to a multiple of divis_by, 'sintel' splits the rows,
    every other mode pads the bottom; unpad inverts pad."""
    from dkt_stereo_amd.utils import InputPadder
    h, w = dims
    x = T(_synth.normal((2, 3, h, w), 7, "pad"))
    p = InputPadder(x.shape, mode=mode, divis_by=div)
    # the reference's arithmetic (utils.py:10-15)
    pad_ht = (((h // div) + 1) * div - h) % div
    pad_wd = (((w // div) + 1) * div - w) % div
    want = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2] if mode == "sintel" else \
        [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]
    assert p._pad == want
    y, = p.pad(x)
    assert y.shape[-2] % div == 0 and y.shape[-1] % div == 0
    assert torch.equal(y, torch.nn.functional.pad(x, want, mode="replicate"))
    assert torch.equal(p.unpad(y), x)


def test_gwcnet_state_dict_matches_reference():
    """dkt_stereo_amd.gwcnet.GWCNet carries the reference's parameter names and shapes (strict=True loading)."""
    from dkt_stereo_amd.gwcnet import GWCNet
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "MANIFEST.json")))["gwcnet_state_dict"]
    got = {k: list(v.shape) for k, v in GWCNet().state_dict().items()}
    assert got == want


def _write_kitti_like(d, seed, H, W, shift):
    """A synthetic pair + ground truth in the KITTI on-disk formats: 8-bit RGB PNGs, 16-bit disparity PNG (x256)."""
    from PIL import Image
    i1, i2 = _synth.image_pair(seed, 1, H, W, shift)
    paths = [os.path.join(d, n) for n in ("left_10.png", "right_10.png", "disp_10.png")]
    for p, im in zip(paths, (i1, i2)):
        Image.fromarray(np.floor(im[0]).clip(0, 255).astype(np.uint8).transpose(1, 2, 0)).save(p)
    disp = _synth.uniform((H, W), 1.0, 60.0, seed, "gt")
    disp[::5, ::7] = 0.0                                       # KITTI marks invalid pixels with 0
    Image.fromarray(np.round(disp * 256.0).astype(np.uint16)).save(paths[2])
    return paths, np.round(disp * 256.0) / 256.0


def test_evaluate_chain_on_cpu(tmp_path):
    """load_sample follows core/stereo_datasets.py (flow = -disparity, KITTI validity = disparity > 0),
    validate pads to /32, un-pads and aggregates EPE / D1 as tools/evaluate_stereo.py:149-166 does."""
    from dkt_stereo_amd import evaluate
    paths, disp = _write_kitti_like(str(tmp_path), 31, 37, 83, 12)
    img1, img2, flow_gt, valid = evaluate.load_sample(*paths)
    assert img1.shape == (3, 37, 83) and img1.dtype == torch.float32 and float(img1.max()) <= 255.0
    assert np.array_equal(flow_gt[0].numpy(), -disp.astype(np.float32))
    assert np.array_equal(valid.numpy(), (disp > 0).astype(np.float32))
    seen = {}

    class Fake(torch.nn.Module):
        """Predicts the ground truth shifted by +1 px (and +5 px in one corner): known EPE and D1."""

        def forward(self, a, b, iters=0, test_mode=False):
            seen["shape"] = tuple(a.shape)
            assert test_mode and a.shape[-2] % 32 == 0 and a.shape[-1] % 32 == 0
            pad = evaluate.InputPadder((1, 3, 37, 83), divis_by=32)
            gt = pad.pad(flow_gt[None])[0] + 1.0
            gt[..., :16 + pad._pad[2], :16 + pad._pad[0]] += 4.0
            return None, gt

    res = evaluate.validate(Fake(), [tuple(paths)], iters=3, device="cpu", keep=True)
    assert seen["shape"] == (1, 3, 64, 96)
    err = np.ones((37, 83))
    err[:16, :16] = 5.0
    val = (disp > 0) & (-disp > -192) & (-disp < 0)
    assert abs(res["epe"] - err[val].mean()) < 1e-6
    assert abs(res["d1"] - 100.0 * (err[val] > 3.0).mean()) < 1e-9
    assert res["predictions"][0].shape == (1, 37, 83) and res["n"] == 1
    # several samples: the evaluator averages per-image EPE but pools the outlier masks
    res2 = evaluate.validate(Fake(), [tuple(paths), (img1, img2, flow_gt, valid)], iters=3, device="cpu")
    assert abs(res2["epe"] - res["epe"]) < 1e-6 and res2["n"] == 2


def test_capture_guard_excludes_other_forwards():
    """update.GPU_GUARD: forwards hold it shared (concurrently), a capture upgrades to exclusive -- no other
    thread is inside a forward while it runs -- and nothing deadlocks when several threads upgrade."""
    import time
    from dkt_stereo_amd.update import _CaptureGuard
    g = _CaptureGuard()
    inside, overlap, order = [0], [0], []
    lock = threading.Lock()

    def worker(i):
        for _ in range(20):
            with g.shared():
                with lock:
                    inside[0] += 1
                with g.shared():                    # re-entrant
                    pass
                if i % 2 == 0:
                    with lock:
                        inside[0] -= 1              # hand the slot back for the upgrade, as exclusive() does
                    with g.exclusive():
                        with lock:
                            if inside[0] != 0:
                                overlap[0] += 1
                            order.append(i)
                        time.sleep(0.0005)
                    with lock:
                        inside[0] += 1
                else:
                    time.sleep(0.0002)
                with lock:
                    inside[0] -= 1

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert not any(t.is_alive() for t in ts), "deadlock"
    assert overlap[0] == 0 and len(order) == 60
    with g.exclusive():                             # also usable without holding it shared
        pass


def test_data_parallel_replica_knows_its_master_and_the_copy_follows_it():
    """torch.nn.parallel.replicate calls _replicate_for_data_parallel on the master (tools/ft_dkt.py:119-125): the replica
    keeps a weak reference to it (its test_mode forward on a GPU goes to the master's persistent per-device copy,
    raft_stereo._Shadow); the copy's refresh logic -- weights when the master's versions change, training flags, instance
    switches -- is plain host code and is checked here on the CPU."""
    import gc
    from dkt_stereo_amd import raft_stereo as rs
    model = rs.RAFTStereo()
    model.eval()
    rep = model._replicate_for_data_parallel()
    assert rep._is_replica and rep._dp_master() is model and not getattr(model, "_is_replica", False)
    sh = rs._Shadow.__new__(rs._Shadow)                 # (no worker thread, no device: _sync alone)
    sh.device, sh.model, sh.fingerprint = torch.device("cpu"), None, None
    sh._sync(model)
    assert sh.model is not model and not sh.model.training
    for a, b in zip(sh._tensors(model), sh._tensors(sh.model)):
        assert a.data_ptr() != b.data_ptr() and torch.equal(a, b)
    fp = sh.fingerprint
    sh._sync(model)
    assert sh.fingerprint == fp                         # nothing changed: nothing copied
    with torch.no_grad():
        model.update_block.flow_head.conv2.weight.mul_(2.0)
    model.use_hip_graph = False
    model.train()
    model.freeze_bn()
    v = sh.model.update_block.flow_head.conv2.weight._version
    sh._sync(model)
    assert sh.fingerprint != fp and sh.model.update_block.flow_head.conv2.weight._version > v    # written in place: caches notice
    assert torch.equal(sh.model.update_block.flow_head.conv2.weight, model.update_block.flow_head.conv2.weight)
    assert sh.model.use_hip_graph is False and sh.model.training
    assert all(not m.training for m in sh.model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    del rep, model
    gc.collect()
