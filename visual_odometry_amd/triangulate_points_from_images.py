"""TriangulatePointsFromTwoImages (reference: src/triangulate_points_from_images.py:7-41, driven by
src/main_triangulate.py): two images -> ORB -> match -> essential matrix -> pose -> triangulated points.

The reference's version targets an ImagePair API it has since deleted (ImagePair(detector).set_images(...)
.standard_pipeline()) and cannot run at HEAD; this is the working equivalent on the current ImagePair surface
(the order of the old pipeline, image_pair.py:214-220 = the order of visual_slam.py:294-298).  The plotly /
imshow display steps are out of scope."""
from __future__ import annotations

import numpy as np

from .frame_generator import FrameGenerator
from .image_and_keypoints import ImageAndKeypoints
from .image_pair import ImagePair


def _imread_bgr(filename):
    from PIL import Image          # cv2.imread stand-in (decoding is frame ingest, outside the hot path)
    rgb = np.asarray(Image.open(filename).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class TriangulatePointsFromTwoImages:
    def __init__(self, camera_matrix=None, verbose=False):
        self.camera_matrix = camera_matrix
        self.verbose = verbose

    def load_images(self, filename_one, filename_two):
        self.img1 = _imread_bgr(filename_one)
        self.img2 = _imread_bgr(filename_two)

    def run(self, filename_one, filename_two):
        self.load_images(filename_one, filename_two)
        return self.run_arrays(self.img1, self.img2)

    def run_arrays(self, img1, img2):
        image1 = ImageAndKeypoints("ORB"); image1.set_image(img1)
        image2 = ImageAndKeypoints("ORB"); image2.set_image(img2)
        K = image1.cameraMatrix if self.camera_matrix is None else np.asarray(self.camera_matrix, dtype=np.float64)
        gen = FrameGenerator(image1.detector)
        frame1, frame2 = gen.make_frame(image1.image), gen.make_frame(image2.image)
        pair = ImagePair(frame1, frame2, image1.bf, K)
        pair.verbose = self.verbose
        pair.match_features()
        inliers = pair.determine_essential_matrix(pair.filtered_matches)
        pair.estimate_camera_movement(inliers)
        pair.reconstruct_3d_points(inliers)
        self.pair = pair
        self.visualization = pair.visualize_matches(inliers)
        return pair
