"""Frame record (reference: src/frame.py:4-18): image, id, keypoints, descriptors, features."""


class Frame:
    def __init__(self, image=None):
        self.image = image
        self.id = None
        self.keypoints = None
        self.descriptors = None
        self.features = None

    def __repr__(self):
        return repr("Frame %d" % self.id)
