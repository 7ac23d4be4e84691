"""Diagnostic: cycles k_ransac spends per phase (needs a library whose geom_kernels.hip was built with -DVO_EXP_TIMING, which
repurposes the count fields of vo_pair_result as clock64() deltas; never use such a build for anything else)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
C = 64
seq = synth.sequence(9, 1280, 720, cache_dir="/tmp")
order = [(i % 16 if i % 16 < 9 else 16 - i % 16) for i in range(C + 1)]
frames = seq["frames"][order]
fe = FrontEnd(720, 1280, C + 1, C, nfeatures=2000)
fe.upload(frames); fe.detect(0, C + 1)
pairs = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
res, _ = fe.run_pairs(pairs, seq["K"])
# VO_EXP_TIMING build: n_kp1 = sampling, n_kp2 = solve, n_match = scoring (first round), n_good = rest, reserved = iters
print("cycles: sample %.0f solve %.0f score %.0f rest %.0f iters %.1f inl %.0f" % (res["n_kp1"].mean(), res["n_kp2"].mean(), res["n_match"].mean(), res["n_good"].mean(), res["reserved"].mean(), res["n_inl"].mean()))
