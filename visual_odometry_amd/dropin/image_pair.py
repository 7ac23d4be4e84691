"""Drop-in module name for the reference's `from image_pair import ImagePair`."""
from visual_odometry_amd.image_pair import ImagePair, isRotationMatrix, rotationMatrixToEulerAngles  # noqa: F401
