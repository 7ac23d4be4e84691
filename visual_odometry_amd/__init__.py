"""visual_odometry_amd — MI355X (gfx950) per-frame-pair visual-odometry front end.

Host code is Python; every arithmetic stage is a hand-written HIP kernel in libvo_hip.so reached through a
ctypes C ABI (include/vo_hip.h).  The classes mirror the reference's call surface
(Samirez/Visual_odometry: src/frame.py, src/frame_generator.py, src/image_pair.py, src/initials.py) and
the cv2 objects it injects (ORB detector, BFMatcher).  Importing the package does not touch the GPU;
the library is loaded on first use and its absence is an error (no CPU fallback).
"""
from .types import KeyPoint, DMatch  # noqa: F401
from .initials import Feature, Match, Match3D, MatchWithMap  # noqa: F401
from .frame import Frame  # noqa: F401
from .frame_generator import FrameGenerator  # noqa: F401

__all__ = ["KeyPoint", "DMatch", "Feature", "Match", "Match3D", "MatchWithMap", "Frame", "FrameGenerator",
           "OrbDetector", "ORB_create", "SiftDetector", "SIFT_create", "HammingMatcher", "L2Matcher", "BFMatcher", "NORM_HAMMING", "NORM_L2", "ImagePair", "ImageAndKeypoints",
           "TriangulatePointsFromTwoImages", "FrontEnd"]


def __getattr__(name):
    # lazily import the modules that bind libvo_hip.so
    if name in ("OrbDetector", "ORB_create", "SiftDetector", "SIFT_create"):
        from . import detector
        return getattr(detector, name)
    if name in ("HammingMatcher", "L2Matcher", "BFMatcher", "NORM_HAMMING", "NORM_L2"):
        from . import matcher
        return getattr(matcher, name)
    if name == "ImagePair":
        from .image_pair import ImagePair
        return ImagePair
    if name == "ImageAndKeypoints":
        from .image_and_keypoints import ImageAndKeypoints
        return ImageAndKeypoints
    if name == "TriangulatePointsFromTwoImages":
        from .triangulate_points_from_images import TriangulatePointsFromTwoImages
        return TriangulatePointsFromTwoImages
    if name == "FrontEnd":
        from .frontend import FrontEnd
        return FrontEnd
    raise AttributeError(name)
