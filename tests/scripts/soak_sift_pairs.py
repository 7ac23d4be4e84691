#!/usr/bin/env python3
"""Checker script (GPU box): the reference's LIVE configuration, frame-batched (FrontEnd(detector="sift"): SIFT -> int8 matrix-core L2
matcher -> findEssentialMat -> recoverPose -> points) on rendered flights of random size, step, yaw and SIFT parameters against the
oracle: every frame's keypoints and descriptors, every pair's match list (indices and float distances), mask, E, R | t bit for bit
(tests/test_gpu_sift_batch.py's checks on fresh inputs).
    python tests/scripts/soak_sift_pairs.py [--seconds 300] [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=300); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from visual_odometry_amd import _lib, synth, frontend as F
    from oracle import oracle as O
    from test_gpu_sift_batch import _check_pairs, _same_features
    O.set_dk_early_exit(True) if hasattr(O, "set_dk_early_exit") else None      # the product's default root-finder rule on both sides
    rng = np.random.default_rng(a.seed)
    t0 = tick = time.time(); nfr = npairs = 0
    try:
        while time.time() - t0 < a.seconds:
            if time.time() - tick > 30: tick = time.time(); print(f"... {nfr} frames, {npairs} pairs identical so far", flush=True)
            w, h = int(rng.integers(160, 700)), int(rng.integers(120, 500))
            n = int(rng.integers(2, 5))
            kw = dict(nOctaveLayers=int(rng.integers(2, 5)), contrastThreshold=float(rng.uniform(0.02, 0.06)), edgeThreshold=float(rng.uniform(6, 14)),
                      sigma=float(rng.uniform(1.3, 2.0)))
            seq = synth.sequence(n, w, h, step=float(rng.uniform(0.3, 2.0)), yaw_deg=float(rng.uniform(0, 1.5)), seed=int(rng.integers(0, 1 << 30)), workers=8)
            ctx = _lib.Context(0)
            fe = F.FrontEnd(h, w, max_frames=n, max_pairs=n, detector="sift", kp_cap=4096, ctx=ctx, **kw)
            fe.upload(seq["frames"]); fe.detect(0, n)
            feats = [fe.features(s) for s in range(n)]
            tag = dict(w=w, h=h, n=n, seed=a.seed, frames_so_far=nfr, **kw)
            if any(f["truncated"] for f in feats): ctx.close(); continue
            try:
                for s in range(n):
                    _same_features(feats[s], O.sift_detect_and_compute(seq["frames"][s], n_layers=kw["nOctaveLayers"], contrast_threshold=kw["contrastThreshold"],
                                                                       edge_threshold=kw["edgeThreshold"], sigma=kw["sigma"]))
                pairs = [[i, i + 1] for i in range(n - 1)]
                _check_pairs(O, fe, feats, pairs, seq["K"], F.MATCH_CROSSCHECK, 2)
            except AssertionError as e:
                print("MISMATCH", tag, repr(e)[:300]); sys.exit(1)
            nfr += n; npairs += n - 1
            ctx.close()
    finally:
        O.set_dk_early_exit(False) if hasattr(O, "set_dk_early_exit") else None
    print(json.dumps({"frames": nfr, "pairs": npairs, "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
