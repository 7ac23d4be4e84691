"""RANSAC iteration counts / matches / inliers of 16 consecutive pairs of the bench sequence."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
seq = synth.sequence(17, 1280, 720, cache_dir="/tmp")
fe = FrontEnd(720, 1280, max_frames=17, max_pairs=16, nfeatures=2000)
fe.upload(seq["frames"]); fe.detect(0, 17)
pairs = np.stack([np.arange(16), np.arange(16) + 1], 1)
res, _ = fe.run_pairs(pairs, seq["K"])
print("iters", res["ransac_iters"].tolist())
print("match", res["n_match"].tolist())
print("inl", res["n_inl"].tolist())
print("good", res["n_good"].tolist())
