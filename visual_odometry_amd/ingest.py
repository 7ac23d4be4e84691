"""Frame ingest on the MI355X: the resize in front of the path (src/visual_slam.py:346-352).

    img = cv2.imread(filename)                       # JPEG decode: host work, not rebuilt
    img = cv2.resize(img, (int(w * s), int(h * s)))  # -> ingest.resize(img, (int(w * s), int(h * s)))

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
