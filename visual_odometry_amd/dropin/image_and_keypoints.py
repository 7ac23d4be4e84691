"""Drop-in module name for the reference's `from image_and_keypoints import ImageAndKeypoints`."""
from visual_odometry_amd.image_and_keypoints import ImageAndKeypoints  # noqa: F401
