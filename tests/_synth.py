"""Seeded synthetic inputs and weights shared by make_golden.py, the tests,
bench.py and smoke().  Everything derives from numpy PCG64 streams so the same
arrays are regenerated bit-identically anywhere (fixtures store outputs only).
Distributions follow SURVEY.md section 8d.
"""
import zlib

import numpy as np


def rng(*key):
    """Independent PCG64 stream per (seed, name...) key."""
    words = []
    for k in key:
        words.append(zlib.crc32(k.encode()) if isinstance(k, str) else int(k) & 0xFFFFFFFF)
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(words)))


def normal(shape, *key, scale=1.0):
    return (rng(*key).standard_normal(shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)


def uniform(shape, lo, hi, *key):
    return rng(*key).uniform(lo, hi, shape).astype(np.float32)


def fmap_pair(seed, B, C, H, W, W2=None):
    W2 = W if W2 is None else W2
    return normal((B, C, H, W), seed, "fmap1"), normal((B, C, H, W2), seed, "fmap2")


def coords(seed, B, H, W, spread=60.0):
    """(B,2,H,W): x = w - U[0,spread) (includes negative / out-of-range taps), y = h."""
    c = np.zeros((B, 2, H, W), np.float32)
    c[:, 0] = np.arange(W, dtype=np.float32)[None, None, :] - uniform((B, H, W), 0.0, spread, seed, "coords")
    c[:, 1] = np.arange(H, dtype=np.float32)[None, :, None]
    return c


def coords_hard(seed, B, H, W):
    """Edge cases: exact integers, far out of range on both sides, huge values."""
    c = coords(seed, B, H, W, spread=float(W))
    x = c[:, 0]
    x[..., 0::7] = np.round(x[..., 0::7])          # exactly integral -> floor ties
    x[..., 3::11] = -1000.0                        # far left
    x[..., 5::13] = 3.0 * W + 0.25                 # far right
    x[..., 1::17] = W - 1.0                        # last valid column
    x[..., 2::19] = -0.5
    return c


def image_pair(seed, B, H, W, shift=12):
    """image1 ~ U[0,255); image2 = roll(image1, -shift, W) + N(0, 2^2)."""
    i1 = uniform((B, 3, H, W), 0.0, 255.0, seed, "image1")
    i2 = (np.roll(i1, -shift, axis=3) + normal((B, 3, H, W), seed, "image2", scale=2.0)).astype(np.float32)
    return i1, i2


def state_dict(shapes, seed, dtype_of=None):
    """Deterministic weights for a {name: shape} map (any nn.Module.state_dict()).
    Conv weights: encoders (cnet./fnet.) kaiming-normal fan_out like
    core/extractor.py:150-157, everything else nn.Conv2d's default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)).  Norm layers get non-trivial affine
    parameters and running statistics so that they are exercised."""
    out = {}
    norm_prefixes = {k[:-len(".running_mean")] for k in shapes if k.endswith(".running_mean")}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        canon = name.replace(".downsample.1.", ".norm3.")   # same tensor upstream
        prefix, leaf = canon.rsplit(".", 1)
        is_norm = (name.rsplit(".", 1)[0] in norm_prefixes) or (prefix in norm_prefixes)
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, np.int64)
        elif leaf == "running_mean":
            out[name] = normal(shape, seed, canon, scale=0.05)
        elif leaf == "running_var":
            out[name] = uniform(shape, 0.8, 1.2, seed, canon)
        elif is_norm and leaf == "weight":
            out[name] = uniform(shape, 0.9, 1.1, seed, canon)
        elif is_norm and leaf == "bias":
            out[name] = normal(shape, seed, canon, scale=0.02)
        elif leaf == "weight" and len(shape) == 4:
            cout, cin, kh, kw = shape
            if name.startswith(("cnet.", "fnet.")):
                out[name] = normal(shape, seed, canon, scale=float(np.sqrt(2.0 / (cout * kh * kw))))
            else:
                b = 1.0 / float(np.sqrt(cin * kh * kw))
                out[name] = uniform(shape, -b, b, seed, canon)
        elif leaf == "weight" and len(shape) == 5:
            # Conv3d / ConvTranspose3d of the 3-D aggregation networks: fan-out normal like gwc_main.py:221-223
            out[name] = normal(shape, seed, canon, scale=float(np.sqrt(2.0 / (shape[0] * shape[2] * shape[3] * shape[4]))))
        elif leaf == "bias" and len(shape) == 1:
            w = shapes.get(name.rsplit(".", 1)[0] + ".weight")
            fan_in = int(np.prod(tuple(w)[1:])) if w is not None else shape[0]
            b = 1.0 / float(np.sqrt(fan_in))
            out[name] = uniform(shape, -b, b, seed, canon)
        else:
            raise KeyError("no recipe for parameter %s %s" % (name, shape))
    return out


def torch_state_dict(shapes, seed):
    import torch
    return {k: torch.from_numpy(v) for k, v in state_dict(shapes, seed).items()}


def shapes_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: tuple(v.shape) for k, v in sd.items()}
