"""Host-side checks that need no GPU: state-dict compatibility with the
reference (keys + shapes recorded in golden/MANIFEST.json from the reference's own
modules), the oracle firewall, and the C-ABI surface."""
import ctypes
import json
import os
import re
from types import SimpleNamespace

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def update_block_shapes(igev, cfg):
    from dkt_stereo_amd.update import BasicMultiUpdateBlock, BasicMultiUpdateBlockIGEV
    cls = BasicMultiUpdateBlockIGEV if igev else BasicMultiUpdateBlock
    blk = cls(SimpleNamespace(**cfg), hidden_dims=cfg["hidden_dims"])
    return {k: tuple(v.shape) for k, v in blk.state_dict().items()}


def manifest():
    return json.load(open(os.path.join(HERE, "golden", "MANIFEST.json")))


def test_raft_state_dict_matches_reference():
    from dkt_stereo_amd.raft_stereo import RAFTStereo
    mine = {k: list(v.shape) for k, v in RAFTStereo().state_dict().items()}
    ref = manifest()["raft_state_dict"]
    assert sorted(mine) == sorted(ref)
    assert mine == ref


@pytest.mark.parametrize("name,igev,n", [("raft3", False, 3), ("raft2", False, 2), ("raft1", False, 1), ("igev3", True, 3)])
def test_update_block_keys_match_reference(name, igev, n):
    cfg = dict(corr_levels=2 if igev else 4, corr_radius=4, n_downsample=2, n_gru_layers=n,
               hidden_dims=[128, 128, 128], slow_fast_gru=False)
    assert sorted(update_block_shapes(igev, cfg)) == manifest()["state_dict_keys"][name]


def test_pins_recorded_within_bounds():
    pins = manifest()["pins"]
    assert len(pins) > 100
    for name, p in pins.items():
        assert p["max_abs"] <= p["bound"], name


def test_product_never_touches_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "dkt_stereo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "dkt_oracle" not in src and "torch_oracle" not in src, f


def test_no_cpu_fallback():
    """Product operators must fail loudly off-GPU instead of computing on the host."""
    from dkt_stereo_amd import _ffi
    from dkt_stereo_amd.corr import CorrBlock1D
    from dkt_stereo_amd.submodule import build_gwc_volume
    f = torch.zeros(1, 4, 2, 8)
    with torch.no_grad():
        with pytest.raises(_ffi.DktError):
            CorrBlock1D(f, f, num_levels=1, radius=1)
        with pytest.raises(_ffi.DktError):
            build_gwc_volume(f, f, 2, 2)


# ---- C ABI -----------------------------------------------------------------------
def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dktstereo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dkt_[a-z0-9_]+)\s*\(", hdr)))


def test_abi_exports_every_declared_symbol():
    from dkt_stereo_amd import _ffi
    lib = _ffi.lib()
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), n
    assert set(_ffi.SIGNATURES) | {"dkt_version", "dkt_strerror"} == set(names)
    assert lib.dkt_version() == 1
    assert lib.dkt_strerror(0) == b"ok"
    assert b"null" in lib.dkt_strerror(-1)


def test_abi_argument_errors_before_launch():
    """Host-side validation returns negative codes without touching a device."""
    from dkt_stereo_amd import _ffi
    lib = _ffi.lib()
    null = ctypes.c_void_p(0)
    nullpp = ctypes.cast(null, ctypes.POINTER(ctypes.c_void_p))
    assert lib.dkt_corr1d_build(null, null, nullpp, 1, 1, 1, 1, 1, 1, 1.0, -1, null) == -1
    assert lib.dkt_pool_w(null, null, 1, 4, -1, null) == -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    arr = (ctypes.c_void_p * 8)(*([p.value] * 8))
    pp = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
    assert lib.dkt_corr1d_build(p, p, pp, 0, 1, 1, 1, 1, 1, 1.0, -1, null) == -2      # B = 0
    assert lib.dkt_corr1d_build(p, p, pp, 1, 1, 1, 4, 4, 9, 1.0, -1, null) == -3      # L > 8
    assert lib.dkt_corr1d_build(p, p, pp, 1, 1, 1, 4, 4, 4, 1.0, -1, null) == -3      # 4 >> 3 == 0
    assert lib.dkt_corr1d_lookup(pp, p, 8, p, 1, 1, 4, 4, 1, 9, -1, null) == -4       # radius > 8
    assert lib.dkt_gwc_volume(p, p, p, 1, 6, 1, 4, 2, 4, 64, -1, null) == -5          # 6 % 4
    assert lib.dkt_gwc_volume(p, p, p, 1, 8192, 1, 16, 2, 1, 32, -1, null) == -7     # one group's target row > 160 KB of LDS
    assert lib.dkt_concat_volume(p, p, p, 1, 2, 1, 4, 2, 1, 3, -1, null) == -2        # bstride too small
