"""FrameGenerator (reference: src/frame_generator.py:5-38): runs the injected detector once per image,
numbers frames from 0 and tags every feature with (frame_id, index)."""
from .frame import Frame
from .initials import Feature


class FrameGenerator:
    def __init__(self, detector):
        self.next_image_counter = 0
        self.detector = detector

    def make_frame(self, image) -> Frame:
        frame = Frame(image)
        frame.id = self.next_image_counter
        self.next_image_counter += 1
        frame.keypoints, frame.descriptors = self.detector.detectAndCompute(frame.image, None)
        frame.features = [Feature(kp, desc, (frame.id, idx))
                          for idx, (kp, desc) in enumerate(zip(frame.keypoints, frame.descriptors))]
        return frame
