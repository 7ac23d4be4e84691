"""ImagePair with the reference's call surface (src/image_pair.py:222-354), geometry on HIP kernels.

Same constructor, methods and attributes that src/visual_slam.py reads (raw_matches, filtered_matches,
essential_matrix, R, t (3x1), relative_pose, null_projection_matrix, projection_matrix,
points3d_reconstr (4xM, w = 1), matches_with_3d_information); cv2.findEssentialMat / recoverPose /
triangulatePoints are replaced by visual_odometry_amd.geometry.
"""
from __future__ import annotations

import numpy as np

from . import geometry
from .initials import Match, Match3D

DISTANCE_GATE = 1130          # image_pair.py:258 (a SIFT-L2 gate; never binds for Hamming <= 256)
RANSAC_CONFIDENCE = 0.99      # image_pair.py:278
RANSAC_THRESHOLD_PX = 1       # image_pair.py:279


def isRotationMatrix(R):
    """image_pair.py:12-18."""
    return np.linalg.norm(np.identity(3, dtype=R.dtype) - np.dot(np.transpose(R), R)) < 1e-6


def rotationMatrixToEulerAngles(R):
    """image_pair.py:21-43: XYZ Euler angles of a rotation matrix."""
    assert isRotationMatrix(R)
    sy = np.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])
    return np.array([np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), 0.0])


class ImagePair:
    verbose = True      # the reference prints from estimate_camera_movement / reconstruct_3d_points

    def __init__(self, frame1, frame2, matcher, camera_matrix):
        self.frame1 = frame1
        self.frame2 = frame2
        self.matcher = matcher
        self.camera_matrix = camera_matrix

    def match_features(self):
        f1, f2 = self.frame1.features, self.frame2.features
        self.raw_matches = [
            Match(f1[m.queryIdx].feature_id, f2[m.trainIdx].feature_id,
                  f1[m.queryIdx].keypoint.pt, f2[m.trainIdx].keypoint.pt,
                  f1[m.queryIdx].descriptor, f2[m.trainIdx].descriptor,
                  m.distance, np.random.random(3))
            for m in self.matcher.match(self.frame1.descriptors, self.frame2.descriptors)]
        self.filtered_matches = [m for m in self.raw_matches if m.distance < DISTANCE_GATE]

    def get_image_points(self, matches):
        p1 = np.array([m.keypoint1 for m in matches], dtype=np.float64).reshape(-1, 2)
        p2 = np.array([m.keypoint2 for m in matches], dtype=np.float64).reshape(-1, 2)
        return p1, p2

    def determine_essential_matrix(self, matches):
        p1, p2 = self.get_image_points(matches)
        E, mask = geometry.findEssentialMat(p1, p2, self.camera_matrix, geometry.FM_RANSAC,
                                            RANSAC_CONFIDENCE, RANSAC_THRESHOLD_PX)
        if E is None:
            # cv2 returns None here and the reference then dies with AttributeError (image_pair.py:289)
            raise ValueError(f"essential matrix needs >= 5 usable matches, got {len(matches)}")
        self.essential_matrix = E
        return [m for m, keep in zip(matches, mask.ravel() == 1) if keep]

    def estimate_camera_movement(self, matches):
        p1, p2 = self.get_image_points(matches)
        _, self.R, self.t, _ = geometry.recoverPose(self.essential_matrix, p1, p2, self.camera_matrix)
        self.relative_pose = np.eye(4)
        self.relative_pose[:3, :3] = self.R
        self.relative_pose[:3, 3] = self.t.T[0]
        if self.verbose:
            print("relative movement in image pair")
            print(self.relative_pose)

    def reconstruct_3d_points(self, matches, first_projection_matrix=None, second_projection_matrix=None):
        K = self.camera_matrix
        self.null_projection_matrix = K @ np.eye(3, 4)
        self.projection_matrix = K @ np.hstack((self.R.T, -self.R.T @ self.t))
        if first_projection_matrix is not None:
            self.null_projection_matrix = K @ first_projection_matrix
        if second_projection_matrix is not None:
            self.projection_matrix = K @ second_projection_matrix
        p1, p2 = self.get_image_points(matches)
        X = geometry.triangulatePoints(self.projection_matrix, self.null_projection_matrix, p1.T, p2.T)
        X /= X[3, :]
        self.points3d_reconstr = X
        self.matches_with_3d_information = [
            Match3D(m.featureid1, m.featureid2, m.keypoint1, m.keypoint2, m.descriptor1, m.descriptor2,
                    m.distance, m.color, (X[0, i], X[1, i], X[2, i]))
            for i, m in enumerate(matches)]
        if self.verbose:
            print("Reconstructed points")
            print(X.transpose().shape)
            print(X.transpose())

    def visualize_matches(self, matches):
        """Side-by-side image with one line per match (numpy rasteriser; the reference uses cv2.line)."""
        h, w = self.frame1.image.shape[:2]
        a, b = self.frame1.image, self.frame2.image
        if a.ndim == 2:
            a, b = np.stack([a] * 3, axis=2), np.stack([b] * 3, axis=2)
        vis = np.concatenate((a, b), axis=1).copy()
        for m in matches:
            x0, y0 = int(m.keypoint1[0]), int(m.keypoint1[1])
            x1, y1 = int(m.keypoint2[0] + w), int(m.keypoint2[1])
            n = max(abs(x1 - x0), abs(y1 - y0), 1)
            xs = np.clip(np.rint(np.linspace(x0, x1, n + 1)).astype(int), 0, vis.shape[1] - 1)
            ys = np.clip(np.rint(np.linspace(y0, y1, n + 1)).astype(int), 0, vis.shape[0] - 1)
            vis[ys, xs] = np.clip(np.asarray(m.color) * 256, 0, 255).astype(vis.dtype)
        return vis
