#!/usr/bin/env python3
"""Per-pair comparison of the HIP path against the CPU oracle on a seeded synthetic sequence: prints n_match /
n_inl / n_good of both, the RANSAC iteration count and |d[R|t]|_F (0.0 = bit identical); exit code 1 if any pair
differs.  Needs an MI355X."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402
from visual_odometry_amd.frontend import FrontEnd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--nlevels", type=int, default=8)
    ap.add_argument("--stride", type=int, default=1, help="pair (i, i+stride)")
    ap.add_argument("--step", type=float, default=1.0, help="camera advance per frame")
    ap.add_argument("--match-mode", type=int, default=0)
    ap.add_argument("--ratio", type=float, default=0.8)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    seq = synth.sequence(a.frames, a.width, a.height, step=a.step, cache_dir="/tmp")
    fe = FrontEnd(a.height, a.width, a.frames, a.frames, nfeatures=a.nfeatures, nlevels=a.nlevels)
    fe.upload(seq["frames"]); fe.detect(0, a.frames)
    pairs = [[i, i + a.stride] for i in range(a.frames - a.stride)]
    opts = fe.make_opts(match_mode=a.match_mode, ratio=a.ratio, want_points=True)
    res, X = fe.run_pairs(pairs, seq["K"], opts)
    p = O.orb_params(nfeatures=a.nfeatures, nlevels=a.nlevels)
    bad = 0
    for k, (i, j) in enumerate(pairs):
        r = O.pair(seq["frames"][i], seq["frames"][j], p, seq["K"], match_mode=a.match_mode, ratio=a.ratio)
        g = res[k]
        if r["rc"] != 0 or g["status"] != 0:
            same = (r["rc"] != 0) == (g["status"] != 0) and g["n_match"] == r["n_match"]
            print(k, "status", g["status"], "oracle rc", r["rc"], "n_match", g["n_match"], r["n_match"], "" if same else "  <<<<<")
            bad += not same
            continue
        d = np.linalg.norm(np.hstack([g["R"].reshape(3, 3), g["t"].reshape(3, 1)]) - np.hstack([r["R"], r["t"]]))
        n = r["n_inl"]
        dx = np.abs(X[k][:, :n] - r["X"][:, :n]).max() if n else 0.0
        ok = (g["n_match"], g["n_inl"], g["n_good"]) == (r["n_match"], r["n_inl"], r["n_good"]) and d == 0.0 and dx == 0.0
        bad += not ok
        if not a.quiet or not ok:
            print(k, g["n_match"], r["n_match"], g["n_inl"], r["n_inl"], g["n_good"], r["n_good"], g["ransac_iters"],
                  f"dRt {d:.3e} dX {dx:.3e}", "" if ok else "  <<<<<")
    print(f"{len(pairs) - bad}/{len(pairs)} pairs bit-identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
