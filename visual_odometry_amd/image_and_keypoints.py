"""ImageAndKeypoints (reference: src/image_and_keypoints.py:4-51): an image with its detector and matcher — ORB +
NORM_HAMMING + crossCheck for the name "ORB" (:7-9, the only live ORB instantiation in the reference), SIFT + NORM_L2 +
crossCheck for any other name (:10-13) — plus the per-keypoint colour sample of detect_keypoints (:44-51).  Detector and
matcher are the HIP-backed objects."""
from __future__ import annotations

import numpy as np

from .detector import ORB_create, SIFT_create
from .ingest import INTER_AREA, resize
from .matcher import BFMatcher, NORM_HAMMING, NORM_L2


class ImageAndKeypoints:
    def __init__(self, detector_name="ORB"):
        if detector_name == "ORB":
            self.detector = ORB_create()                               # image_and_keypoints.py:8 (defaults: 500 features)
            self.bf = BFMatcher(NORM_HAMMING, crossCheck=True)         # image_and_keypoints.py:9
        else:                                                          # the reference takes SIFT for every other name (:10-13)
            self.detector = SIFT_create()                              # image_and_keypoints.py:12
            self.bf = BFMatcher(NORM_L2, crossCheck=True)              # image_and_keypoints.py:13
        # values from ../input/toys2/calibration.xml, as the reference leaves them (:28-30)
        self.cameraMatrix = np.array([[835.69, 0.0, 1008 / 2 + 61.6], [0.0, 827, 756 / 2 - 9.4], [0.0, 0.0, 1.0]])
        self.distCoeffs = np.array([[0.0097935857180804498, -0.021794052829051412, 0.0046443590741258711,
                                     -0.0045664024579022498, 0.017776502734846815]])
        self.scale_factor = 1
        self.cameraMatrix *= self.scale_factor
        self.cameraMatrix[2, 2] = 1

    def set_image(self, image):
        width = int(image.shape[1] * self.scale_factor)
        height = int(image.shape[0] * self.scale_factor)
        # cv2.resize(image, dim, interpolation=cv2.INTER_AREA)   (image_and_keypoints.py:42; scale_factor is 1 there)
        self.image = resize(image, (width, height), interpolation=INTER_AREA)

    def detect_keypoints(self):
        self.keypoints, self.descriptors = self.detector.detectAndCompute(self.image, None)
        self.kp_colors = [self.image[int(kp.pt[1]), int(kp.pt[0])] for kp in self.keypoints]
