"""Value types with the attribute names the reference reads from cv2 objects."""
from __future__ import annotations


class KeyPoint:
    """cv2.KeyPoint stand-in: the reference reads `.pt` (image_pair.py:246-247, image_and_keypoints.py:49)."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size=31.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size = float(size)
        self.angle = float(angle)
        self.response = float(response)
        self.octave = int(octave)
        self.class_id = int(class_id)

    def __repr__(self):
        return f"KeyPoint(pt={self.pt}, size={self.size:.2f}, angle={self.angle:.2f}, octave={self.octave})"


class DMatch:
    """cv2.DMatch stand-in: `.queryIdx`, `.trainIdx`, `.distance` (image_pair.py:244-250)."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return f"DMatch({self.queryIdx}->{self.trainIdx}, d={self.distance:g})"
