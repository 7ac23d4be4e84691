"""Drop-in module name for the reference's `from initials import *` (schemas + numpy only)."""
import collections  # noqa: F401
import math  # noqa: F401
import time  # noqa: F401
import glob  # noqa: F401

import numpy as np  # noqa: F401

from visual_odometry_amd.initials import Feature, Match, Match3D, MatchWithMap  # noqa: F401
