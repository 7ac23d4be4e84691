"""Upload + three detections of N frames: a small target for rocprofv3 --pmc runs on single kernels."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
seq = synth.sequence(17, 1280, 720, cache_dir="/tmp")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
fe = FrontEnd(720, 1280, max_frames=n, max_pairs=4, nfeatures=2000)
order = [i % 17 for i in range(n)]
fe.upload(seq["frames"][order])
for _ in range(3):
    fe.detect(0, n)
