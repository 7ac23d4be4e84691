"""Record schemas shared by the front end, field for field those of the reference's prelude
(src/initials.py:16-28), without its heavyweight imports (cv2, g2o, OpenGL, plotly ...)."""
import collections

import numpy as np

np.set_printoptions(precision=4, suppress=True)

Feature = collections.namedtuple("Feature", ["keypoint", "descriptor", "feature_id"])
Match = collections.namedtuple("Match", ["featureid1", "featureid2", "keypoint1", "keypoint2",
                                         "descriptor1", "descriptor2", "distance", "color"])
Match3D = collections.namedtuple("Match3D", ["featureid1", "featureid2", "keypoint1", "keypoint2",
                                             "descriptor1", "descriptor2", "distance", "color", "point"])
MatchWithMap = collections.namedtuple("MatchWithMap", ["featureid1", "featureid2", "imagecoord", "mapcoord",
                                                       "descriptor1", "descriptor2", "distance"])
