"""Per-pair comparison of the HIP path against the CPU oracle on a KITTI-shaped synthetic sequence: prints
n_match / n_inl / n_good of both, the RANSAC iteration count and |d[R|t]|_F (0.0 = bit identical). Needs an MI355X."""
import sys, numpy as np
sys.path.insert(0,'.')
from oracle import oracle as O
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
seq = synth.sequence(24, 1241, 376, cache_dir="/tmp")
fe = FrontEnd(376, 1241, 24, 23, nfeatures=2000)
fe.upload(seq["frames"]); fe.detect(0, 24)
pairs = [[i, i + 1] for i in range(23)]
res, _ = fe.run_pairs(pairs, seq["K"]); res = res.copy()
p = O.orb_params(nfeatures=2000)
for k,(i,j) in enumerate(pairs):
    r = O.pair(seq["frames"][i], seq["frames"][j], p, seq["K"], want_points=False)
    g = res[k]
    d = np.linalg.norm(np.hstack([g["R"].reshape(3,3), g["t"].reshape(3,1)]) - np.hstack([r["R"], r["t"]]))
    flag = "" if d < 1e-6 else "  <<<<<"
    print(k, g["n_match"], r["n_match"], g["n_inl"], r["n_inl"], g["n_good"], r["n_good"], g["ransac_iters"], f"{d:.3e}", flag)
    if d > 1e-6:
        print("   E diff", np.abs(g["E"].reshape(3,3)-r["E"]).max(), np.abs(g["E"].reshape(3,3)+r["E"]).max())
