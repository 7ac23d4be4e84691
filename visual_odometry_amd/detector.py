"""ORB detector object with cv2's call surface (`detectAndCompute(image, mask)`), backed by HIP kernels.

Stands in for `cv2.ORB_create(...)` as injected into FrameGenerator (reference: src/visual_slam.py:16,21,
src/frame_generator.py:25-26, src/image_and_keypoints.py:8,46).
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from .types import KeyPoint

HARRIS_SCORE, FAST_SCORE = 0, 1


def make_params(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
                scoreType=HARRIS_SCORE, patchSize=31, fastThreshold=20) -> _lib.OrbParams:
    return _lib.OrbParams(int(nfeatures), float(scaleFactor), int(nlevels), int(edgeThreshold), int(firstLevel),
                          int(WTA_K), int(scoreType), int(patchSize), int(fastThreshold))


class OrbDetector:
    def __init__(self, nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
                 scoreType=HARRIS_SCORE, patchSize=31, fastThreshold=20, ctx: _lib.Context | None = None,
                 keypoint_order: str = "cv2"):
        self.params = make_params(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType,
                                  patchSize, fastThreshold)
        self._ctx = ctx
        if keypoint_order not in ("canonical", "cv2"):
            raise ValueError("keypoint_order must be 'canonical' or 'cv2'")
        self.keypoint_order = keypoint_order   # 'cv2': the list order (hence every keypoint / match index) cv2.ORB returns
        self.truncated = False            # set by detectAndCompute: the last call hit a list capacity

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def capacity(self):
        n = self.params.nfeatures
        return n + max(n // 8, 256) + 8

    def detect_arrays(self, image):
        """Array form: dict(xy [N,2] f32, size, angle, response [N] f32, octave [N] i32, desc [N,32] u8)."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim not in (2, 3):
            raise TypeError("image must be a uint8 array of shape HxW, HxWx3 (BGR) or HxWx4 (BGRA)")
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        cap = self.capacity()
        xy = np.empty((cap, 2), np.float32); size = np.empty(cap, np.float32); ang = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32); octv = np.empty(cap, np.int32); desc = np.empty((cap, 32), np.uint8)
        n = C.c_int32(0)
        ctx = self.ctx
        ctx.set_keypoint_order(self.keypoint_order)
        rc = ctx.check(ctx.lib.vo_orb_detect_and_compute(
            ctx.handle, img.ctypes.data, h, w, ch, img.strides[0], C.addressof(self.params),
            xy.ctypes.data, size.ctypes.data, ang.ctypes.data, resp.ctypes.data, octv.ctypes.data,
            desc.ctypes.data, cap, C.addressof(n)))
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=ang[:k].copy(), response=resp[:k].copy(),
                    octave=octv[:k].copy(), desc=desc[:k].copy(), truncated=(rc == _lib.VO_WARN_CAPACITY))

    def detectAndCompute(self, image, mask=None):
        if mask is not None:
            raise NotImplementedError("detection masks are not supported (the reference always passes None)")
        a = self.detect_arrays(image)
        # cv2 keeps every tie with the n-th response; this build's lists have a fixed capacity
        self.truncated = a["truncated"]
        if a["truncated"]:
            warnings.warn("ORB candidate / keypoint capacity reached: the keypoint list was truncated in canonical order "
                          "(cv2 would have kept every tie)", RuntimeWarning, stacklevel=2)
        kps = tuple(KeyPoint(x, y, s, an, r, o) for (x, y), s, an, r, o in
                    zip(a["xy"].tolist(), a["size"].tolist(), a["angle"].tolist(), a["response"].tolist(),
                        a["octave"].tolist()))
        return kps, a["desc"]

    # stage outputs used by the parity tests
    def _stage(self, fn_name, image):
        img = np.ascontiguousarray(image)
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        ctx = self.ctx
        nbytes = int(ctx.lib.vo_packed_pyramid_bytes(int(h), int(w), C.addressof(self.params)))
        if nbytes < 0:
            raise ValueError("invalid ORB parameters for this image size")
        out = np.empty(nbytes, np.uint8)
        ctx.check(getattr(ctx.lib, fn_name)(ctx.handle, img.ctypes.data, h, w, ch, img.strides[0],
                                            C.addressof(self.params), out.ctypes.data))
        return out

    def stage_levels(self, fn_name, image, level_sizes):
        flat = self._stage(fn_name, image)
        levels, off = [], 0
        for (lw, lh) in level_sizes:
            levels.append(flat[off:off + lw * lh].reshape(lh, lw).copy())
            off += lw * lh
        return levels


def ORB_create(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
               scoreType=HARRIS_SCORE, patchSize=31, fastThreshold=20, keypoint_order="cv2") -> OrbDetector:
    """cv2.ORB_create look-alike.  keypoint_order='cv2' (default) returns the keypoints in cv2's own list order (so keypoint and
    match indices are cv2's); 'canonical' = (level, y, x) order, the same set, 0.36 ms per 257 frames cheaper."""
    return OrbDetector(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize,
                       fastThreshold, keypoint_order=keypoint_order)


class SiftDetector:
    """cv2.SIFT_create(...) with cv2's call surface — the detector the reference actually runs (src/visual_slam.py:17),
    injected into FrameGenerator (:21) and paired with BFMatcher(NORM_L2, crossCheck=True) (:19, matcher.L2Matcher).
    detectAndCompute(image, None) -> (keypoints, descriptors [N, 128] float32 with values 0..255), keypoints in the order
    cv2 returns them, .octave packed as cv2 packs it (octave | layer << 8 | round((xi + 0.5) * 255) << 16)."""

    def __init__(self, nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10, sigma=1.6, ctx: _lib.Context | None = None):
        self.params = _lib.SiftParams(int(nfeatures), int(nOctaveLayers), float(contrastThreshold), float(edgeThreshold), float(sigma))
        self._ctx = ctx
        self.truncated = False

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def detect_arrays(self, image, cap=1 << 16):
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim not in (2, 3):
            raise TypeError("image must be a uint8 array of shape HxW, HxWx3 (BGR) or HxWx4 (BGRA)")
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        ctx = self.ctx
        while True:
            xy = np.empty((cap, 2), np.float32); size = np.empty(cap, np.float32); ang = np.empty(cap, np.float32)
            resp = np.empty(cap, np.float32); octv = np.empty(cap, np.int32); desc = np.empty((cap, 128), np.float32)
            n = C.c_int32(0)
            rc = ctx.lib.vo_sift_detect_and_compute(ctx.handle, img.ctypes.data, h, w, ch, img.strides[0], C.addressof(self.params),
                                                    xy.ctypes.data, size.ctypes.data, ang.ctypes.data, resp.ctypes.data, octv.ctypes.data,
                                                    desc.ctypes.data, cap, C.addressof(n))
            if rc == _lib.VO_ERR_UNSUPPORTED:
                raise NotImplementedError(ctx.last_error())
            ctx.check(rc)
            if n.value <= cap or cap >= (1 << 18):
                break
            cap = 1 << 18                                      # the device lists' own capacity
        k = min(n.value, cap)
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=ang[:k].copy(), response=resp[:k].copy(), octave=octv[:k].copy(),
                    desc=desc[:k].copy(), truncated=(rc == _lib.VO_WARN_CAPACITY))

    def detectAndCompute(self, image, mask=None):
        if mask is not None:
            raise NotImplementedError("detection masks are not supported (the reference always passes None)")
        a = self.detect_arrays(image)
        self.truncated = a["truncated"]
        if a["truncated"]:
            warnings.warn("SIFT candidate / keypoint capacity reached: the keypoint list was truncated", RuntimeWarning, stacklevel=2)
        kps = tuple(KeyPoint(x, y, s, an, r, o) for (x, y), s, an, r, o in
                    zip(a["xy"].tolist(), a["size"].tolist(), a["angle"].tolist(), a["response"].tolist(), a["octave"].tolist()))
        return kps, a["desc"]


def SIFT_create(nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10, sigma=1.6) -> SiftDetector:
    """cv2.SIFT_create look-alike (src/visual_slam.py:17)."""
    return SiftDetector(nfeatures, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)
