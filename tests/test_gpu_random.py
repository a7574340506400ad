"""Randomised parity (-m gpu): hypothesis draws shapes / parameters, the HIP path is compared with
the C oracle (bit-exact for the sampler, pooling and volume kernels) or an fp64 convolution.
Small shapes, a few dozen examples each: ragged widths, W1 != W2, single rows, odd channel counts,
batch > 1, every radius / level combination the ABI accepts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import _synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SET = dict(deadline=None, max_examples=60, suppress_health_check=list(HealthCheck), derandomize=True)


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def same(a, b):
    """Bit-for-bit equality; a width-1 pyramid level makes the reference divide by W-1 = 0, and the
    NaNs it then produces must appear in the same places."""
    a = a.cpu().numpy() if torch.is_tensor(a) else a
    b = b.cpu().numpy() if torch.is_tensor(b) else b
    return np.array_equal(a, b, equal_nan=True)


@settings(**SET)
@given(B=st.integers(1, 3), C=st.integers(1, 40), H=st.integers(1, 5), W1=st.integers(2, 70), dW=st.integers(-8, 12),
       L=st.integers(1, 4), r=st.integers(0, 5), seed=st.integers(0, 10 ** 6))
@torch.no_grad()
def test_corr_block_random(c_oracle, B, C, H, W1, dW, L, r, seed):
    from dkt_stereo_amd.corr import CorrBlock1D, _lookup
    W2 = max(W1 + dW, 1 << (L - 1))
    f1 = _synth.normal((B, C, H, W1), seed, "f1")
    f2 = _synth.normal((B, C, H, W2), seed, "f2")
    x = _synth.uniform((B, 1, H, W1), -6.0, W2 + 6.0, seed, "x")
    x[..., ::3] = np.round(x[..., ::3])
    coords = np.concatenate([x, np.zeros_like(x)], 1)
    blk = CorrBlock1D(G(f1), G(f2), num_levels=L, radius=r)
    pyr = [p.view(p.shape[0], -1).cpu().numpy() for p in blk.corr_pyramid]
    want0 = c_oracle.corr1d_build(f1, f2, 1)[0]
    scale = max(float(np.abs(want0).max()), 1.0)
    assert float(np.abs(pyr[0] - want0).max()) <= 4e-6 * scale
    own = c_oracle.pool_pyramid(pyr[0], L)
    for i in range(1, L):
        assert np.array_equal(pyr[i], own[i])                              # pooling: bit exact
    out = blk(G(coords))
    assert same(out, c_oracle.corr1d_lookup(pyr, coords, r))                  # sampler: bit exact
    assert same(out, _lookup(blk.corr_pyramid, G(coords), r, W2))             # skew == row layout


@settings(**SET)
@given(B=st.integers(1, 2), G_=st.integers(1, 6), cpg=st.integers(1, 24), H=st.integers(1, 4), W=st.integers(1, 40),
       D=st.integers(1, 20), seed=st.integers(0, 10 ** 6))
@torch.no_grad()
def test_volumes_random(c_oracle, B, G_, cpg, H, W, D, seed):
    from dkt_stereo_amd.submodule import build_concat_volume, build_concat_volume_igev, build_gwc_volume, build_gwc_volume_norm
    C = G_ * cpg
    a = _synth.normal((B, C, H, W), seed, "a")
    b = _synth.normal((B, C, H, W), seed, "b")
    from dkt_stereo_amd import submodule as sm
    with sm.gwc_mode("exact"):
        assert np.array_equal(build_gwc_volume(G(a), G(b), D, G_).cpu().numpy(), c_oracle.gwc_volume(a, b, D, G_))
        assert np.array_equal(build_gwc_volume_norm(G(a), G(b), D, G_).cpu().numpy(), c_oracle.gwc_volume_norm(a, b, D, G_))
    want = c_oracle.gwc_volume(a, b, D, G_)               # default mode (MFMA where the shape allows): one rounding per product
    assert np.abs(build_gwc_volume(G(a), G(b), D, G_).cpu().numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    if C <= 16:
        assert np.array_equal(build_concat_volume(G(a), G(b), D).cpu().numpy(), c_oracle.concat_volume(a, b, D, 1))
        assert np.array_equal(build_concat_volume_igev(G(a), G(b), D).cpu().numpy(), c_oracle.concat_volume(a, b, D, 0))


@settings(**SET)
@given(B=st.integers(1, 2), chans=st.lists(st.integers(1, 70), min_size=1, max_size=3), cout=st.integers(1, 200),
       H=st.integers(1, 20), W=st.integers(1, 70), k=st.sampled_from([1, 3]), stride=st.sampled_from([1, 2]),
       relu=st.booleans(), seed=st.integers(0, 10 ** 6))
@torch.no_grad()
def test_conv_random(B, chans, cout, H, W, k, stride, relu, seed):
    from dkt_stereo_amd import conv
    conv.set_backend("f16x3")
    if stride == 2:
        chans = chans[:1]
    cin = sum(chans)
    torch.manual_seed(seed)
    layer = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).to(DEV)
    xs = [G(_synth.normal((B, c, H, W), seed, "x%d" % i, scale=1.5)) for i, c in enumerate(chans)]
    ref = F.conv2d(torch.cat(xs, 1).double(), layer.weight.double(), layer.bias.double(), stride=stride, padding=k // 2)
    ref = ref.clamp_min(0) if relu else ref
    got = conv.conv2d(xs if len(xs) > 1 else xs[0], layer, relu=relu)
    assert got.shape == ref.shape
    assert float((got.double() - ref).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max()))


@settings(**SET)
@given(B=st.integers(1, 2), C=st.integers(1, 12), H=st.integers(1, 4), W=st.integers(4, 60), L=st.integers(1, 3),
       S=st.sampled_from([1, 3, 5, 9]), Gn=st.integers(1, 3), ds=st.sampled_from([2, 3]), seed=st.integers(0, 10 ** 6))
@torch.no_grad()
def test_pcv_random(c_oracle, B, C, H, W, L, S, Gn, ds, seed):
    from dkt_stereo_amd.pcvnet_corr import CorrBlock1D
    f = 4 if ds == 2 else 2
    if W // f ** (L - 1) < 1:
        L = 1
    f1 = _synth.normal((B, C, H, W), seed, "f1")
    f2 = _synth.normal((B, C, H, W), seed, "f2")
    coords = _synth.uniform((B, Gn, H, W), -4.0, W + 4.0, seed, "c")
    sigma = _synth.uniform((B, Gn, H, W), 0.1, 4.0, seed, "s")
    blk = CorrBlock1D(G(f1), G(f2), sample_num=S, num_levels=L, downsample=ds)
    pyr = [p.view(p.shape[0], -1).cpu().numpy() for p in blk.corr_pyramid]
    own = c_oracle.pcv_pyramid(pyr[0], L, f)
    for i in range(1, L):
        assert np.array_equal(pyr[i], own[i])
    assert same(blk(G(coords), G(sigma)), c_oracle.pcv_lookup(pyr, coords, sigma, S, f))
