"""cv2.findEssentialMat / recoverPose / triangulatePoints look-alikes backed by HIP kernels
(reference call sites: src/image_pair.py:280-286, :304-308, :332-336)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

RANSAC = FM_RANSAC = 8
OPENCV_RNG_SEED = 0xFFFFFFFFFFFFFFFF


def _pts(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    if p.ndim == 3 and p.shape[1] == 1:
        p = p.reshape(-1, 2)
    if p.ndim != 2 or p.shape[1] != 2:
        raise ValueError("points must be an M x 2 array")
    return np.ascontiguousarray(p)


def findEssentialMat(points1, points2, cameraMatrix, method=RANSAC, prob=0.999, threshold=1.0, maxIters=1000,
                     seed=OPENCV_RNG_SEED, ctx=None):
    """Returns (E 3x3 float64, mask Mx1 uint8); (None, None) for fewer than 5 points, as cv2 does."""
    if method != RANSAC:
        raise NotImplementedError("only RANSAC (cv2.FM_RANSAC) is implemented")
    p1, p2 = _pts(points1), _pts(points2)
    if len(p1) != len(p2):
        raise ValueError("point sets differ in length")
    K = np.ascontiguousarray(cameraMatrix, dtype=np.float64).reshape(3, 3)
    M = len(p1)
    if M < 5:
        return None, None
    ctx = ctx or _lib.default_context()
    E = np.zeros((10, 9)); mask = np.zeros(M, np.uint8)
    ninl = C.c_int32(0); nmod = C.c_int32(0)
    rc = ctx.lib.vo_find_essential_ransac(ctx.handle, p1.ctypes.data, p2.ctypes.data, M, K.ctypes.data, float(prob),
                                          float(threshold), int(maxIters), int(seed), E.ctypes.data, mask.ctypes.data,
                                          C.addressof(ninl), C.addressof(nmod))
    if rc == _lib.VO_ERR_NO_MODEL:
        return None, None
    ctx.check(rc)
    return E[:nmod.value].reshape(-1, 3).copy(), mask.reshape(-1, 1)


def recoverPose(E, points1, points2, cameraMatrix, distanceThresh=50.0, ctx=None):
    """Returns (n_good, R 3x3, t 3x1, mask Mx1 with 0/255)."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    if E.shape != (3, 3):
        raise ValueError("E must be 3x3")
    p1, p2 = _pts(points1), _pts(points2)
    K = np.ascontiguousarray(cameraMatrix, dtype=np.float64).reshape(3, 3)
    M = len(p1)
    ctx = ctx or _lib.default_context()
    R = np.zeros((3, 3)); t = np.zeros((3, 1)); mask = np.zeros(max(M, 1), np.uint8); ng = C.c_int32(0)
    ctx.check(ctx.lib.vo_recover_pose(ctx.handle, E.ctypes.data, p1.ctypes.data, p2.ctypes.data, M, K.ctypes.data,
                                      float(distanceThresh), R.ctypes.data, t.ctypes.data, mask.ctypes.data,
                                      C.addressof(ng)))
    return ng.value, R, t, mask[:M].reshape(-1, 1)


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2, ctx=None):
    """Returns the 4 x M homogeneous points (not normalised), float64."""
    P1 = np.ascontiguousarray(projMatr1, dtype=np.float64).reshape(3, 4)
    P2 = np.ascontiguousarray(projMatr2, dtype=np.float64).reshape(3, 4)
    x1 = np.ascontiguousarray(projPoints1, dtype=np.float64)
    x2 = np.ascontiguousarray(projPoints2, dtype=np.float64)
    if x1.ndim != 2 or x1.shape[0] != 2 or x1.shape != x2.shape:
        raise ValueError("projPoints must be 2 x M arrays of equal shape")
    M = x1.shape[1]
    X = np.zeros((4, M))
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.vo_triangulate(ctx.handle, P1.ctypes.data, P2.ctypes.data, x1.ctypes.data, x2.ctypes.data, M,
                                     X.ctypes.data))
    return X


def five_point(x1, x2, ctx=None):
    """All essential matrices through 5 normalised correspondences (stage test hook)."""
    x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(5, 2)
    x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(5, 2)
    E = np.zeros((10, 9)); n = C.c_int32(0)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.vo_stage_five_point(ctx.handle, x1.ctypes.data, x2.ctypes.data, E.ctypes.data, C.addressof(n)))
    return E[:n.value].reshape(-1, 3, 3).copy()


SOLVEPNP_ITERATIVE = 0


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, useExtrinsicGuess=False, iterationsCount=100,
                   reprojectionError=8.0, confidence=0.99, flags=SOLVEPNP_ITERATIVE, seed=OPENCV_RNG_SEED, ctx=None):
    """cv2.solvePnPRansac as the reference calls it (src/visual_slam.py:231-235: positional objectPoints, imagePoints,
    cameraMatrix, zeros(4)) -> (retval, rvec 3x1, tvec 3x1, inliers n_inl x 1 int32 indices or None).
    Only the reference's configuration is built: zero distortion, no extrinsic guess, SOLVEPNP_ITERATIVE."""
    if distCoeffs is not None and np.any(np.asarray(distCoeffs, np.float64) != 0):
        raise NotImplementedError("only zero distortion coefficients are built (the reference passes np.zeros(4))")
    if useExtrinsicGuess or flags != SOLVEPNP_ITERATIVE:
        raise NotImplementedError("only cv2.solvePnPRansac's defaults (no extrinsic guess, SOLVEPNP_ITERATIVE) are built")
    obj = np.ascontiguousarray(np.asarray(objectPoints, np.float64).reshape(-1, 3))
    img = np.ascontiguousarray(np.asarray(imagePoints, np.float64).reshape(-1, 2))
    if len(obj) != len(img):
        raise ValueError("objectPoints and imagePoints differ in length")
    K = np.ascontiguousarray(cameraMatrix, np.float64).reshape(3, 3)
    n = len(obj)
    rvec = np.zeros((3, 1)); tvec = np.zeros((3, 1)); mask = np.zeros(max(n, 1), np.uint8); ninl = C.c_int32(0)
    ctx = ctx or _lib.default_context()
    rc = ctx.lib.vo_solve_pnp_ransac(ctx.handle, obj.ctypes.data, img.ctypes.data, n, K.ctypes.data, int(iterationsCount),
                                     float(reprojectionError), float(confidence), int(seed), rvec.ctypes.data, tvec.ctypes.data,
                                     mask.ctypes.data, C.addressof(ninl))
    if rc == _lib.VO_ERR_NO_MODEL:
        return False, rvec, tvec, None                     # cv2: retval False
    ctx.check(rc)                                          # n < 4 raises, as cv2's assertion does
    return True, rvec, tvec, np.nonzero(mask[:n])[0].astype(np.int32).reshape(-1, 1)


def solve_pnp_ransac_batch(objectPoints, imagePoints, offsets, cameraMatrix, iterationsCount=100, reprojectionError=8.0,
                           confidence=0.99, seed=OPENCV_RNG_SEED, ctx=None):
    """B independent solvePnPRansac problems in one launch; problem b owns rows offsets[b]:offsets[b+1].
    Returns (status [B], rvec [B, 3], tvec [B, 3], mask [total] uint8, n_inliers [B])."""
    obj = np.ascontiguousarray(np.asarray(objectPoints, np.float64).reshape(-1, 3))
    img = np.ascontiguousarray(np.asarray(imagePoints, np.float64).reshape(-1, 2))
    off = np.ascontiguousarray(offsets, np.int32)
    B = len(off) - 1
    K = np.ascontiguousarray(cameraMatrix, np.float64).reshape(3, 3)
    rvec = np.zeros((B, 3)); tvec = np.zeros((B, 3)); mask = np.zeros(max(len(obj), 1), np.uint8)
    ninl = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.vo_solve_pnp_ransac_batch(ctx.handle, obj.ctypes.data, img.ctypes.data, off.ctypes.data, B, K.ctypes.data,
                                                int(iterationsCount), float(reprojectionError), float(confidence), int(seed),
                                                rvec.ctypes.data, tvec.ctypes.data, mask.ctypes.data, ninl.ctypes.data,
                                                status.ctypes.data))
    return status, rvec, tvec, mask[:len(obj)], ninl


def Rodrigues(src, ctx=None):
    """cv2.Rodrigues(src) -> (dst, None): 3-vector (3, 3x1 or 1x3) -> 3x3 matrix, 3x3 matrix -> 3x1 vector
    (src/visual_slam.py:243; the Jacobian cv2 also returns is not computed)."""
    a = np.ascontiguousarray(src, np.float64)
    ctx = ctx or _lib.default_context()
    if a.size == 9:
        out = np.zeros((3, 1))
        ctx.check(ctx.lib.vo_rodrigues(ctx.handle, a.ctypes.data, 1, out.ctypes.data))
    elif a.size == 3:
        out = np.zeros((3, 3))
        ctx.check(ctx.lib.vo_rodrigues(ctx.handle, a.ctypes.data, 0, out.ctypes.data))
    else:
        raise ValueError("Rodrigues takes a 3-vector or a 3x3 matrix")
    return out, None
