"""On-disk readers / writers either side of the inference path: the formats
``core/utils/frame_utils.py`` handles for the stereo datasets the reference evaluates on
(``evaluate_stereo.py:83-330``).  Host-side numpy; same function names, argument meaning and
return conventions (``(disp, valid)`` pairs) as the reference.  PIL replaces the reference's
cv2 / imageio calls (neither is needed for these formats).  Not provided: the 48-bit KITTI
*optical-flow* PNGs (readFlowKITTI / writeFlowKITTI need a 16-bit RGB decoder; they are not on
the stereo path).
"""
import json
import re
from os.path import basename, exists, splitext

import numpy as np
from PIL import Image

TAG_CHAR = np.array([202021.25], np.float32)


def readFlow(fn):
    """Middlebury .flo (frame_utils.py:41-60): little-endian, magic 202021.25, then w, h, (h,w,2) fp32."""
    with open(fn, 'rb') as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or 202021.25 != magic[0]:
            print('Magic number incorrect. Invalid .flo file')
            return None
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
        return np.resize(data, (h, w, 2))


def writeFlow(filename, uv, v=None):
    """frame_utils.py:106-136: u and v interleaved per pixel."""
    if v is None:
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    assert u.shape == v.shape
    height, width = u.shape
    with open(filename, 'wb') as f:
        f.write(TAG_CHAR.tobytes())
        np.array(width).astype(np.int32).tofile(f)
        np.array(height).astype(np.int32).tofile(f)
        tmp = np.zeros((height, width * 2))
        tmp[:, np.arange(width) * 2] = u
        tmp[:, np.arange(width) * 2 + 1] = v
        tmp.astype(np.float32).tofile(f)


def readPFM(file):
    """frame_utils.py:62-92: 'PF' (3 channels) / 'Pf' (1), 'W H', scale (negative = little endian),
    rows stored bottom-up."""
    with open(file, 'rb') as f:
        header = f.readline().rstrip()
        if header == b'PF':
            color = True
        elif header == b'Pf':
            color = False
        else:
            raise Exception('Not a PFM file.')
        dim_match = re.match(rb'^(\d+)\s(\d+)\s$', f.readline())
        if not dim_match:
            raise Exception('Malformed PFM header.')
        width, height = map(int, dim_match.groups())
        scale = float(f.readline().rstrip())
        endian = '<' if scale < 0 else '>'
        data = np.fromfile(f, endian + 'f')
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape))


def writePFM(file, array):
    """frame_utils.py:94-104: single-channel, little endian."""
    assert type(file) is str and type(array) is np.ndarray and splitext(file)[1] == ".pfm"
    with open(file, 'wb') as f:
        H, W = array.shape
        for header in ("Pf\n", "%d %d\n" % (W, H), "-1\n"):
            f.write(str.encode(header))
        f.write(np.flip(array, axis=0).astype(np.float32).tobytes())


def readDispKITTI(filename):
    """frame_utils.py:152-155: 16-bit PNG, disparity = value / 256, 0 = invalid."""
    disp = np.array(Image.open(filename)).astype(np.uint16) / 256.0
    valid = disp > 0.0
    return disp, valid


def readDispSintelStereo(file_name):
    """frame_utils.py:158-164."""
    a = np.array(Image.open(file_name))
    d_r, d_g, d_b = np.split(a, axis=2, indices_or_sections=3)
    disp = (d_r * 4 + d_g / (2 ** 6) + d_b / (2 ** 14))[..., 0]
    mask = np.array(Image.open(file_name.replace('disparities', 'occlusions')))
    valid = ((mask == 0) & (disp > 0))
    return disp, valid


def readDispFallingThings(file_name):
    """frame_utils.py:167-174."""
    a = np.array(Image.open(file_name))
    with open('/'.join(file_name.split('/')[:-1] + ['_camera_settings.json']), 'r') as f:
        intrinsics = json.load(f)
    fx = intrinsics['camera_settings'][0]['intrinsic_settings']['fx']
    disp = (fx * 6.0 * 100) / a.astype(np.float32)
    valid = disp > 0
    return disp, valid


def readDispTartanAir(file_name):
    """frame_utils.py:177-181."""
    depth = np.load(file_name)
    disp = 80.0 / depth
    valid = disp > 0
    return disp, valid


def readDispMiddlebury(file_name):
    """frame_utils.py:184-196."""
    if basename(file_name) == 'disp0GT.pfm':
        disp = readPFM(file_name).astype(np.float32)
        assert len(disp.shape) == 2
        nocc_pix = file_name.replace('disp0GT.pfm', 'mask0nocc.png')
        assert exists(nocc_pix)
        nocc_pix = np.array(Image.open(nocc_pix)) == 255
        assert np.any(nocc_pix)
        return disp, nocc_pix
    elif basename(file_name) == 'disp0.pfm':
        disp = readPFM(file_name).astype(np.float32)
        valid = disp < 1e3
        return disp, valid


def read_gen(file_name, pil=False):
    """frame_utils.py:205-224: dispatch on the extension."""
    ext = splitext(file_name)[-1]
    if ext in ('.png', '.jpeg', '.ppm', '.jpg'):
        return Image.open(file_name)
    elif ext in ('.bin', '.raw', '.npy'):
        return np.load(file_name)
    elif ext == '.flo':
        return readFlow(file_name).astype(np.float32)
    elif ext == '.pfm':
        flow = readPFM(file_name).astype(np.float32)
        if len(flow.shape) == 2:
            return flow
        return flow[:, :, :-1]
    return []
