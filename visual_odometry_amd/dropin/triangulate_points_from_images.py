"""Drop-in module name for `from triangulate_points_from_images import TriangulatePointsFromTwoImages`."""
from visual_odometry_amd.triangulate_points_from_images import TriangulatePointsFromTwoImages  # noqa: F401
