"""Frame ingest on the MI355X: the decode and resize in front of the path (src/visual_slam.py:346-352).

    img = cv2.imread(filename)                       # -> ingest.imread(filename)   (baseline JPEG, decoded on the GPU)
    img = cv2.resize(img, (int(w * s), int(h * s)))  # -> ingest.resize(img, (int(w * s), int(h * s)))

imread() / imdecode() mirror cv2.imread(file) / cv2.imdecode(buf, cv2.IMREAD_COLOR) for baseline JPEG files: what cv2 gets
from libjpeg-turbo (integer IDCT, fancy upsampling, B G R), EXIF orientation applied as cv2 does; decode_batch() takes many
files of one size in a single launch sequence.

resize() mirrors cv2.resize(src, dsize) for 8-bit 1/3/4-channel images with the default INTER_LINEAR and with
INTER_AREA (src/image_and_keypoints.py:42) for any reduction (OpenCV's enlarging INTER_AREA is not built).
"""
from __future__ import annotations

import numpy as np

from . import _lib

INTER_LINEAR = 1      # cv2.INTER_LINEAR
INTER_AREA = 3        # cv2.INTER_AREA


def resize(src, dsize, interpolation=INTER_LINEAR, ctx=None):
    """cv2.resize(src, dsize): dsize = (width, height)."""
    img = np.ascontiguousarray(src, dtype=np.uint8)
    if img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] not in (1, 3, 4)):
        raise ValueError("resize takes an 8-bit image with 1, 3 or 4 channels")
    dw, dh = int(dsize[0]), int(dsize[1])
    if dw < 1 or dh < 1:
        raise ValueError("dsize must be positive")
    sh, sw = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    if interpolation not in (INTER_LINEAR, INTER_AREA):
        raise NotImplementedError("only INTER_LINEAR (cv2.resize's default) and INTER_AREA are built")
    if interpolation == INTER_AREA and (dw > sw or dh > sh):
        raise NotImplementedError("INTER_AREA enlargement (a bilinear variant in OpenCV) is not built")
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    ctx = ctx or _lib.default_context()
    fn = ctx.lib.vo_resize_area if interpolation == INTER_AREA else ctx.lib.vo_resize_linear
    ctx.check(fn(ctx.handle, img.ctypes.data, sh, sw, cn, img.strides[0], out.ctypes.data, dh, dw, out.strides[0]))
    return out


# EXIF orientation -> the array operation cv2.imread applies (ExifTransform in OpenCV's loadsave.cpp)
def _apply_orientation(img, o):
    if o == 2: return img[:, ::-1]
    if o == 3: return img[::-1, ::-1]
    if o == 4: return img[::-1]
    if o == 5: return img.transpose(1, 0, 2)
    if o == 6: return img.transpose(1, 0, 2)[:, ::-1]
    if o == 7: return img.transpose(1, 0, 2)[::-1, ::-1]
    if o == 8: return img.transpose(1, 0, 2)[::-1]
    return img


def jpeg_info(buf):
    """(height, width, components, sampling of component 0 as (h << 4) | v, EXIF orientation or 0, decodable here)."""
    b = np.frombuffer(buf, np.uint8)
    v = [_lib.C.c_int32(0) for _ in range(5)]
    rc = _lib.load().vo_jpeg_info(b.ctypes.data, len(b), *[_lib.C.addressof(x) for x in v])
    if rc == _lib.VO_ERR_INVALID:
        raise ValueError("not a JPEG file")
    return tuple(x.value for x in v) + (rc == 0,)


def imdecode(buf, ctx=None, apply_orientation=True):
    """cv2.imdecode(buf, cv2.IMREAD_COLOR) for a JPEG held in memory -> [h, w, 3] uint8, B G R."""
    b = np.frombuffer(buf, np.uint8)
    h, w, _, _, orient, ok = jpeg_info(b)
    if not ok:
        raise NotImplementedError("JPEG frame type outside the baseline decoder (progressive, lossless, arithmetic or 12-bit)")
    out = np.empty((h, w, 3), np.uint8)
    ctx = ctx or _lib.default_context()
    hh = _lib.C.c_int32(0); ww = _lib.C.c_int32(0)
    rc = ctx.lib.vo_jpeg_decode(ctx.handle, b.ctypes.data, len(b), out.ctypes.data, h, w, _lib.C.addressof(hh), _lib.C.addressof(ww))
    if rc == _lib.VO_ERR_UNSUPPORTED:
        raise NotImplementedError(ctx.last_error())
    ctx.check(rc)
    return np.ascontiguousarray(_apply_orientation(out, orient)) if apply_orientation and orient > 1 else out


def imread(filename, ctx=None):
    """cv2.imread(filename) for a .jpg (src/visual_slam.py:346).  Like cv2, returns None when the file cannot be read."""
    try:
        with open(filename, "rb") as f:
            data = f.read()
    except OSError:
        return None
    return imdecode(data, ctx)


class PackedFiles:
    """JPEG files laid back to back in ONE page-locked host buffer (vo_host_alloc): what vo_jpeg_decode_batch /
    vo_frames_ingest_jpeg take, in memory the DMA engines read directly (a pageable blob is staged by the runtime at a
    fraction of the PCIe rate).  Fill it from bytes objects (one memcpy per file) or let a reader write into
    `blob[offsets[k]:offsets[k + 1]]` itself (file.readinto) and skip that copy too."""

    def __init__(self, buffers=None, sizes=None):
        if buffers is not None:
            sizes = [len(b) for b in buffers]
        self.offsets = np.zeros(len(sizes) + 1, np.int64)
        self.offsets[1:] = np.cumsum(sizes)
        self._hold = _lib.PinnedArray((int(self.offsets[-1]) + 16,), np.uint8)
        self.blob = self._hold.array
        if buffers is not None:
            for k, b in enumerate(buffers):
                self.blob[self.offsets[k]:self.offsets[k + 1]] = np.frombuffer(b, np.uint8)

    def __len__(self):
        return len(self.offsets) - 1

    def file(self, k):
        return self.blob[self.offsets[k]:self.offsets[k + 1]]


def _packed(buffers):
    """(blob, offsets, keep-alive) of a PackedFiles or of a sequence of bytes-like objects."""
    if isinstance(buffers, PackedFiles):
        return buffers.blob, buffers.offsets, buffers
    bufs = [b if isinstance(b, (bytes, bytearray, memoryview)) else bytes(b) for b in buffers]
    blob = np.frombuffer(b"".join(bufs), np.uint8) if bufs else np.zeros(0, np.uint8)
    offs = np.zeros(len(bufs) + 1, np.int64); offs[1:] = np.cumsum([len(b) for b in bufs])
    return blob, offs, bufs


def decode_batch(buffers, ctx=None):
    """Many JPEG files of ONE size (a sequence of bytes-like objects or a PackedFiles) -> [F, h, w, 3] uint8 (B G R) with
    one upload and one launch sequence."""
    blob, offs, _keep = _packed(buffers)
    n = len(offs) - 1
    if n == 0:
        return np.empty((0, 0, 0, 3), np.uint8)
    h, w = jpeg_info(blob[offs[0]:offs[1]])[:2]
    out = np.empty((n, h, w, 3), np.uint8)
    ctx = ctx or _lib.default_context()
    rc = ctx.lib.vo_jpeg_decode_batch(ctx.handle, blob.ctypes.data, offs.ctypes.data, n, out.ctypes.data, h, w)
    if rc == _lib.VO_ERR_UNSUPPORTED:
        raise NotImplementedError(ctx.last_error())
    ctx.check(rc)
    return out
