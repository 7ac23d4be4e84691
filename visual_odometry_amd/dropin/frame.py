"""Drop-in module name for the reference's `from frame import Frame` (see README.md here)."""
from visual_odometry_amd.frame import Frame  # noqa: F401
