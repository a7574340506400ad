"""On-disk readers / writers either side of the inference path: the dataset formats
``core/utils/frame_utils.py`` handles for the stereo benchmarks the reference evaluates on
(``evaluate_stereo.py:83-330``).  Host-side numpy + PIL (the reference's cv2 / imageio calls are not
needed for these formats); function names, argument meaning and return conventions -- arrays, or
``(disparity, valid)`` pairs -- are the reference's, so dataset code can import this module instead.
``tests/test_frame_utils.py`` pins every reader on the reference's own output for the same files.

Not provided: the 48-bit KITTI *optical-flow* PNGs (``readFlowKITTI`` / ``writeFlowKITTI`` need a 16-bit
RGB decoder; they are not on the stereo path).
"""
import json
import os
import struct

import numpy as np
from PIL import Image

FLO_MAGIC = 202021.25                     # Middlebury .flo tag ("PIEH" read as a little-endian float)
TAG_CHAR = np.array([FLO_MAGIC], np.float32)


# ----------------------------------------------------------------------------- Middlebury .flo
def readFlow(fn):
    """frame_utils.py:41-60.  Layout: float32 magic, int32 width, int32 height, then height x width
    (u, v) float32 pairs, all little endian.  A wrong magic prints a message and yields None."""
    raw = open(fn, 'rb').read()
    if len(raw) < 4 or struct.unpack('<f', raw[:4])[0] != FLO_MAGIC:
        print('Magic number incorrect. Invalid .flo file')
        return None
    width, height = struct.unpack('<ii', raw[4:12])
    uv = np.frombuffer(raw, dtype='<f4', count=2 * width * height, offset=12)
    return np.resize(uv, (height, width, 2))


def writeFlow(filename, uv, v=None):
    """frame_utils.py:106-136.  ``uv`` is (H, W, 2), or ``uv`` / ``v`` are the two (H, W) planes."""
    if v is None:
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[..., 0], uv[..., 1]
    else:
        u = uv
    assert u.shape == v.shape
    height, width = u.shape
    interleaved = np.stack([u, v], axis=-1).astype('<f4')
    with open(filename, 'wb') as f:
        f.write(struct.pack('<fii', FLO_MAGIC, width, height))
        f.write(interleaved.tobytes())


# ----------------------------------------------------------------------------- PFM
def readPFM(file):
    """frame_utils.py:62-92.  Header lines: ``PF`` (3 channels) or ``Pf`` (1), ``<width> <height>``, a scale
    whose sign gives the byte order (negative = little endian); rows are stored bottom-up."""
    with open(file, 'rb') as f:
        kind = f.readline().rstrip()
        if kind not in (b'PF', b'Pf'):
            raise Exception('Not a PFM file.')
        dims = f.readline().split()
        if len(dims) != 2 or not all(d.isdigit() for d in dims):
            raise Exception('Malformed PFM header.')
        width, height = int(dims[0]), int(dims[1])
        little = float(f.readline().rstrip()) < 0
        samples = np.fromfile(f, '<f' if little else '>f')
    shape = (height, width, 3) if kind == b'PF' else (height, width)
    return np.flipud(samples.reshape(shape))


def writePFM(file, array):
    """frame_utils.py:94-104: one channel, little endian (scale line ``-1``)."""
    assert type(file) is str and type(array) is np.ndarray and os.path.splitext(file)[1] == ".pfm"
    rows, cols = array.shape
    with open(file, 'wb') as f:
        f.write(("Pf\n%d %d\n-1\n" % (cols, rows)).encode())
        f.write(np.ascontiguousarray(array[::-1], dtype=np.float32).tobytes())


# ----------------------------------------------------------------------------- disparity maps
def _with_valid(disp):
    return disp, disp > 0


def readDispKITTI(filename):
    """frame_utils.py:152-155: 16-bit grey PNG holding disparity * 256; 0 marks missing ground truth."""
    return _with_valid(np.asarray(Image.open(filename), dtype=np.uint16) / 256.0)


def readDispSintelStereo(file_name):
    """frame_utils.py:158-164: disparity packed into the three 8-bit channels (R*4 + G/2^6 + B/2^14);
    the occlusion mask lives in the parallel ``occlusions`` directory."""
    rgb = np.array(Image.open(file_name))
    r, g, b = (rgb[..., i:i + 1] for i in range(3))
    disp = (r * 4 + g / (2 ** 6) + b / (2 ** 14))[..., 0]
    occluded = np.array(Image.open(file_name.replace('disparities', 'occlusions')))
    return disp, (occluded == 0) & (disp > 0)


def readDispFallingThings(file_name):
    """frame_utils.py:167-174: depth PNG in 0.01 cm units, baseline 6 cm, focal length from the sequence's
    ``_camera_settings.json``."""
    depth = np.array(Image.open(file_name))
    with open(os.path.join(os.path.dirname(file_name), '_camera_settings.json')) as f:
        fx = json.load(f)['camera_settings'][0]['intrinsic_settings']['fx']
    return _with_valid((fx * 6.0 * 100) / depth.astype(np.float32))


def readDispTartanAir(file_name):
    """frame_utils.py:177-181: ``.npy`` depth in metres, disparity = 80 / depth."""
    return _with_valid(80.0 / np.load(file_name))


def readDispMiddlebury(file_name):
    """frame_utils.py:184-196: ``disp0GT.pfm`` comes with a non-occlusion mask PNG (255 = visible);
    ``disp0.pfm`` marks invalid pixels with inf."""
    name = os.path.basename(file_name)
    if name == 'disp0GT.pfm':
        disp = readPFM(file_name).astype(np.float32)
        assert disp.ndim == 2
        mask_file = file_name.replace('disp0GT.pfm', 'mask0nocc.png')
        assert os.path.exists(mask_file)
        visible = np.array(Image.open(mask_file)) == 255
        assert np.any(visible)
        return disp, visible
    if name == 'disp0.pfm':
        disp = readPFM(file_name).astype(np.float32)
        return disp, disp < 1e3


# ----------------------------------------------------------------------------- dispatch by extension
def _read_pfm_image(path):
    data = readPFM(path).astype(np.float32)
    return data if data.ndim == 2 else data[:, :, :-1]


_READERS = {
    '.png': Image.open, '.jpeg': Image.open, '.ppm': Image.open, '.jpg': Image.open,
    '.bin': np.load, '.raw': np.load, '.npy': np.load,
    '.flo': lambda p: readFlow(p).astype(np.float32),
    '.pfm': _read_pfm_image,
}


def read_gen(file_name, pil=False):
    """frame_utils.py:205-224: images as PIL objects, arrays for everything else, ``[]`` for an unknown
    extension."""
    reader = _READERS.get(os.path.splitext(file_name)[-1])
    return reader(file_name) if reader is not None else []
