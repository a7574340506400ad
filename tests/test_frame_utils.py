"""dkt_stereo_amd/frame_utils.py against the reference's readers (tests/golden/files.npz holds what
core/utils/frame_utils.py returned for the files under tests/golden/files/, written by
make_golden.py with the reference's own writers where it has one)."""
import os

import numpy as np
import pytest

from dkt_stereo_amd import frame_utils as fu

D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "files")


def test_pfm_roundtrip_and_reference(golden, tmp_path):
    g = golden("files")
    a = fu.readPFM(os.path.join(D, "disp.pfm"))
    assert a.dtype == g["pfm"].dtype and np.array_equal(a, g["pfm"])
    c = fu.readPFM(os.path.join(D, "color_be.pfm"))                      # 3 channels, big endian
    assert np.array_equal(c, g["pfm_color"])
    assert np.array_equal(fu.read_gen(os.path.join(D, "color_be.pfm")), g["read_gen_pfm_color"])
    # our writer produces byte-identical files to the reference's
    out = str(tmp_path / "w.pfm")
    fu.writePFM(out, np.ascontiguousarray(a).astype(np.float32))
    assert open(out, "rb").read() == open(os.path.join(D, "disp.pfm"), "rb").read()
    with pytest.raises(Exception):
        bad = tmp_path / "bad.pfm"
        bad.write_bytes(b"P6\n1 1\n-1\n\x00\x00\x00\x00")
        fu.readPFM(str(bad))


def test_flo(golden, tmp_path):
    g = golden("files")
    uv = fu.readFlow(os.path.join(D, "flow.flo"))
    assert np.array_equal(uv, g["flo"])
    out = str(tmp_path / "w.flo")
    fu.writeFlow(out, uv)
    assert open(out, "rb").read() == open(os.path.join(D, "flow.flo"), "rb").read()
    assert np.array_equal(fu.read_gen(os.path.join(D, "flow.flo")), g["flo"].astype(np.float32))
    (tmp_path / "x.flo").write_bytes(np.array([1.0], np.float32).tobytes())
    assert fu.readFlow(str(tmp_path / "x.flo")) is None               # wrong magic: the reference prints and returns None


def test_kitti_sintel_tartan(golden):
    g = golden("files")
    d, v = fu.readDispKITTI(os.path.join(D, "kitti_disp.png"))
    assert d.dtype == np.float64 and np.array_equal(d, g["kitti_disp_expected"])
    assert np.array_equal(v, g["kitti_disp_expected"] > 0)
    d, v = fu.readDispSintelStereo(os.path.join(D, "disparities", "frame.png"))
    assert np.array_equal(d, g["sintel_disp"]) and np.array_equal(v, g["sintel_valid"])
    d, v = fu.readDispTartanAir(os.path.join(D, "depth.npy"))
    assert np.array_equal(d, g["tartan_disp"]) and np.array_equal(v, g["tartan_valid"])
    assert fu.read_gen("nothing.xyz") == []
