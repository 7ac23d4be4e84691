"""Drop-in module name for the reference's `from frame_generator import FrameGenerator`."""
from visual_odometry_amd.frame_generator import FrameGenerator  # noqa: F401
