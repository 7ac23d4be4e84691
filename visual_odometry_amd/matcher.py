"""Brute-force Hamming / L2 matchers with cv2.BFMatcher's call surface (match, knnMatch), backed by HIP kernels.

Stands in for `cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)` as injected into ImagePair
(reference: src/visual_slam.py:18,294, src/image_and_keypoints.py:9, src/image_pair.py:234-236) and adds
knnMatch(k=2) + the ratio rule of src/feature_detection.py:20-26.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .types import DMatch

NORM_HAMMING = 6


def _desc(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2 or a.shape[1] != 32:
        raise ValueError("descriptors must be an N x 32 uint8 array (ORB, 256 bits)")
    return a


class HammingMatcher:
    def __init__(self, crossCheck: bool = False, legacy_crosscheck: bool = False, ctx: _lib.Context | None = None):
        # crossCheck=True is OpenCV 4.x's rule (opencv-python 4.7.0.72 is what the reference locks): mutual nearest
        # neighbours.  legacy_crosscheck=True selects the older batchDistance rule without the forward test.
        self.crossCheck = bool(crossCheck)
        self.legacy_crosscheck = bool(legacy_crosscheck)
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def match_arrays(self, query, train):
        q, t = _desc(query), _desc(train)
        nq = len(q)
        qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
        n = C.c_int32(0)
        mode = 0 if not self.crossCheck else (1 if self.legacy_crosscheck else 2)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_match_hamming(ctx.handle, q.ctypes.data, nq, t.ctypes.data, len(t), mode,
                                           qi.ctypes.data, ti.ctypes.data, d.ctypes.data, C.addressof(n)))
        k = n.value
        return qi[:k].copy(), ti[:k].copy(), d[:k].copy()

    def match(self, queryDescriptors, trainDescriptors):
        qi, ti, d = self.match_arrays(queryDescriptors, trainDescriptors)
        return [DMatch(a, b, c) for a, b, c in zip(qi.tolist(), ti.tolist(), d.tolist())]

    def ratio_match_arrays(self, query, train, ratio):
        q, t = _desc(query), _desc(train)
        nq = len(q)
        qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
        n = C.c_int32(0)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_knn2_ratio_hamming(ctx.handle, q.ctypes.data, nq, t.ctypes.data, len(t), float(ratio),
                                                qi.ctypes.data, ti.ctypes.data, d.ctypes.data, C.addressof(n)))
        k = n.value
        return qi[:k].copy(), ti[:k].copy(), d[:k].copy()

    def ratio_match(self, queryDescriptors, trainDescriptors, ratio=0.75):
        """knnMatch(k=2) followed by `m.distance < ratio * n.distance` (feature_detection.py:24-26)."""
        qi, ti, d = self.ratio_match_arrays(queryDescriptors, trainDescriptors, ratio)
        return [DMatch(a, b, c) for a, b, c in zip(qi.tolist(), ti.tolist(), d.tolist())]

    def knn2_arrays(self, query, train):
        """Both neighbours of every query row: (idx [nq, 2] int32, dist [nq, 2] float32); a missing neighbour is -1 / FLT_MAX."""
        q, t = _desc(query), _desc(train)
        idx = np.empty((max(len(q), 1), 2), np.int32); d = np.empty((max(len(q), 1), 2), np.float32)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_knn2_hamming(ctx.handle, q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data, d.ctypes.data))
        return idx[:len(q)], d[:len(q)]

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        """cv2's matcher.knnMatch(d1, d2, k=2) (src/feature_detection.py:21,90): one list per query row holding its nearest and
        second-nearest train rows as DMatch objects, so that the script's `for m, n in matches:` runs unchanged."""
        return _knn_rows(self.knn2_arrays(queryDescriptors, trainDescriptors), k, self.crossCheck)


class L2Matcher:
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck) on float32 descriptors — the reference's live matcher
    (src/visual_slam.py:19; SIFT rows, 128 floats: detector.SiftDetector).  Takes any float descriptor rows."""

    def __init__(self, crossCheck: bool = False, legacy_crosscheck: bool = False, ctx: _lib.Context | None = None):
        self.crossCheck = bool(crossCheck)
        self.legacy_crosscheck = bool(legacy_crosscheck)
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        return self._ctx

    def match_arrays(self, query, train):
        q = np.ascontiguousarray(query, dtype=np.float32); t = np.ascontiguousarray(train, dtype=np.float32)
        if q.ndim != 2 or t.ndim != 2 or (len(q) and len(t) and q.shape[1] != t.shape[1]):
            raise ValueError("descriptors must be N x dim float32 arrays of equal dim")
        nq = len(q)
        dim = q.shape[1] if nq else (t.shape[1] if len(t) else 1)
        qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
        n = C.c_int32(0)
        mode = 0 if not self.crossCheck else (1 if self.legacy_crosscheck else 2)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_match_l2(ctx.handle, q.ctypes.data, nq, t.ctypes.data, len(t), int(dim), mode,
                                      qi.ctypes.data, ti.ctypes.data, d.ctypes.data, C.addressof(n)))
        k = n.value
        return qi[:k].copy(), ti[:k].copy(), d[:k].copy()

    def match(self, queryDescriptors, trainDescriptors):
        qi, ti, d = self.match_arrays(queryDescriptors, trainDescriptors)
        return [DMatch(a, b, c) for a, b, c in zip(qi.tolist(), ti.tolist(), d.tolist())]

    def knn2_arrays(self, query, train):
        """Both neighbours of every query row: (idx [nq, 2] int32, dist [nq, 2] float32); a missing neighbour is -1 / FLT_MAX."""
        q, t, dim = self._rows(query, train)
        idx = np.empty((max(len(q), 1), 2), np.int32); d = np.empty((max(len(q), 1), 2), np.float32)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_knn2_l2(ctx.handle, q.ctypes.data, len(q), t.ctypes.data, len(t), dim, idx.ctypes.data, d.ctypes.data))
        return idx[:len(q)], d[:len(q)]

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        """cv2.BFMatcher(cv2.NORM_L2).knnMatch(d1, d2, k=2): what src/feature_detection.py:21 calls on SIFT descriptors."""
        return _knn_rows(self.knn2_arrays(queryDescriptors, trainDescriptors), k, self.crossCheck)

    def ratio_match_arrays(self, query, train, ratio):
        q, t, dim = self._rows(query, train)
        nq = len(q)
        qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
        n = C.c_int32(0)
        ctx = self.ctx
        ctx.check(ctx.lib.vo_knn2_ratio_l2(ctx.handle, q.ctypes.data, nq, t.ctypes.data, len(t), dim, float(ratio),
                                           qi.ctypes.data, ti.ctypes.data, d.ctypes.data, C.addressof(n)))
        k = n.value
        return qi[:k].copy(), ti[:k].copy(), d[:k].copy()

    def ratio_match(self, queryDescriptors, trainDescriptors, ratio=0.75):
        """knnMatch(k=2) followed by `m.distance < ratio * n.distance` (feature_detection.py:24-26) in one call."""
        qi, ti, d = self.ratio_match_arrays(queryDescriptors, trainDescriptors, ratio)
        return [DMatch(a, b, c) for a, b, c in zip(qi.tolist(), ti.tolist(), d.tolist())]

    @staticmethod
    def _rows(query, train):
        q = np.ascontiguousarray(query, dtype=np.float32); t = np.ascontiguousarray(train, dtype=np.float32)
        if q.ndim != 2 or t.ndim != 2 or (len(q) and len(t) and q.shape[1] != t.shape[1]):
            raise ValueError("descriptors must be N x dim float32 arrays of equal dim")
        return q, t, int(q.shape[1] if len(q) else (t.shape[1] if len(t) else 1))


def _knn_rows(arrays, k, cross_check):
    """[idx, dist] [nq, 2] -> cv2's list of per-query DMatch lists.  cv2 leaves out neighbours that do not exist (fewer than k
    train rows) and refuses k > 1 on a cross-checking matcher."""
    if cross_check and k != 1:
        raise ValueError("knnMatch with k > 1 needs crossCheck=False (cv2 asserts knn == 1 || !crossCheck)")
    if k not in (1, 2):
        raise NotImplementedError("knnMatch is built for k = 1 and k = 2 (the reference calls it with k=2)")
    idx, dist = arrays
    out = []
    for q, (ii, dd) in enumerate(zip(idx.tolist(), dist.tolist())):
        out.append([DMatch(q, t, d) for t, d in list(zip(ii, dd))[:k] if t >= 0])
    return out


NORM_L2 = 4


def BFMatcher(normType=NORM_HAMMING, crossCheck=False):
    """cv2.BFMatcher look-alike: NORM_HAMMING (ORB, the north-star path) or NORM_L2 (float descriptors)."""
    if normType == NORM_L2:
        return L2Matcher(crossCheck=crossCheck)
    if normType != NORM_HAMMING:
        raise NotImplementedError("only NORM_HAMMING and NORM_L2 are implemented")
    return HammingMatcher(crossCheck=crossCheck)
