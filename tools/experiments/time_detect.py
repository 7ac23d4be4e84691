"""Per-stage HIP-event times of the detection chain alone (257 frames 1280x720, 2000 features), three repeats."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
seq = synth.sequence(17, 1280, 720, cache_dir="/tmp")
n = 257
fe = FrontEnd(720, 1280, max_frames=n, max_pairs=4, nfeatures=2000)
order = [(i % 32 if i % 32 < 17 else 32 - i % 32) for i in range(n)]
fe.upload(seq["frames"][order])
for _ in range(2): fe.detect(0, n)
out = []
for rep in range(3):
    fe.profile(True)
    for _ in range(4): fe.detect(0, n)
    p = fe.profile_read(); fe.profile(False)
    out.append(" ".join(f"{k[:6]}={v[0]/v[1]:.3f}" for k, v in p.items()))
print(sys.argv[1] if len(sys.argv) > 1 else "", "stage ms/launch:", out[-1], " kp0:", len(fe.features(0)["xy"]))
