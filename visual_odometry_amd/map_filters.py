"""Reprojection-error filter over all map observations (reference: src/map.py:46-94).

The reference walks Python lists of Observation / TrackedCamera / TrackedPoint objects and multiplies 4x4
matrices per observation; here the same quantities go to the GPU as arrays (one lane per observation).  The
object-level helpers accept the reference's record types unchanged (duck typed: .camera_id/.pose(),
.point_id/.point, .camera_id/.point_id/.image_coordinates)."""
from __future__ import annotations

import numpy as np

from . import _lib


def reprojection_sqerr(poses, points, obs_cam, obs_pt, obs_xy, camera_matrix, threshold=100.0, ctx=None):
    """Arrays in, (sqerr [N] float64, keep [N] bool) out. obs_cam / obs_pt index rows of poses / points."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    xy = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(camera_matrix, np.float64).reshape(3, 3)
    n = len(oc)
    if not (len(op) == n == len(xy)):
        raise ValueError("observation arrays differ in length")
    err = np.zeros(n); keep = np.zeros(n, np.uint8)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.vo_reprojection_filter(ctx.handle, poses.ctypes.data, len(poses), points.ctypes.data, len(points),
                                             oc.ctypes.data, op.ctypes.data, xy.ctypes.data, n, K.ctypes.data,
                                             float(threshold), err.ctypes.data, keep.ctypes.data))
    return err, keep.astype(bool)


def _arrays(cameras, points, observations):
    cam_row = {c.camera_id: i for i, c in enumerate(cameras)}
    pt_row = {p.point_id: i for i, p in enumerate(points)}
    poses = np.stack([c.pose() for c in cameras]) if cameras else np.zeros((0, 4, 4))
    pts = np.array([p.point for p in points], dtype=np.float64).reshape(-1, 3)
    oc = np.array([cam_row[o.camera_id] for o in observations], np.int32)
    op = np.array([pt_row[o.point_id] for o in observations], np.int32)
    xy = np.array([o.image_coordinates for o in observations], dtype=np.float64).reshape(-1, 2)
    return poses, pts, oc, op, xy


def remove_observations_with_reprojection_errors_above_threshold(cameras, points, observations, camera_matrix,
                                                                 threshold=100, ctx=None):
    """map.py:46-68 — returns the observations whose squared reprojection error is below the threshold."""
    if not observations:
        return []
    _, keep = reprojection_sqerr(*_arrays(cameras, points, observations), camera_matrix, threshold, ctx)
    return [o for o, k in zip(observations, keep) if k]


def calculate_reprojection_error(cameras, points, observations, camera_matrix, ctx=None):
    """map.py:70-94 — total squared reprojection error, summed in observation order."""
    if not observations:
        return 0.0
    err, _ = reprojection_sqerr(*_arrays(cameras, points, observations), camera_matrix, np.inf, ctx)
    total = 0.0
    for e in err.tolist():
        total += e
    return total


def feature_tracks(n_frames, cap, pair_frames, matches, ctx=None):
    """update_feature_mapper + track_feature_back_in_time (visual_slam.py:183-188, :94-99) for every feature at once.

    pair_frames: [P, 2] frame ids (f1, f2); matches: list of P (q_idx, t_idx) int arrays — the matches with 3-D
    information of each pair, in the order the reference would process the pairs.  Returns (root_frame, root_idx,
    hops), each [n_frames, cap]: feature (f, i) traces back to (root_frame[f, i], root_idx[f, i])."""
    pf = np.ascontiguousarray(pair_frames, np.int32).reshape(-1, 2)
    P = len(pf)
    off = np.zeros(P + 1, np.int32)
    for p, (q, t) in enumerate(matches):
        if len(q) != len(t):
            raise ValueError("q and t differ in length")
        off[p + 1] = off[p] + len(q)
    mq = np.ascontiguousarray(np.concatenate([np.asarray(q, np.int32) for q, _ in matches]) if P else np.zeros(0, np.int32))
    mt = np.ascontiguousarray(np.concatenate([np.asarray(t, np.int32) for _, t in matches]) if P else np.zeros(0, np.int32))
    rf = np.zeros((n_frames, cap), np.int32); ri = np.zeros((n_frames, cap), np.int32); hops = np.zeros((n_frames, cap), np.int32)
    ctx = ctx or _lib.default_context()
    ctx.check(ctx.lib.vo_feature_tracks(ctx.handle, int(n_frames), int(cap), pf.ctypes.data, off.ctypes.data, mq.ctypes.data,
                                        mt.ctypes.data, P, rf.ctypes.data, ri.ctypes.data, hops.ctypes.data))
    return rf, ri, hops
