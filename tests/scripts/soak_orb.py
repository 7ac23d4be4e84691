#!/usr/bin/env python3
"""Checker script (GPU box): random problems through the hot path and the oracle, bit for bit —
  * the ORB detector (image class, size, feature count, pyramid depth, FAST threshold, score type: tests/test_gpu_orb.py's
    fuzz generator with fresh seeds) in cv2's keypoint order,
  * the two-view stage (findEssentialMat's masks and E, recoverPose's R | t and counts) in the `opencv300` root-finder mode on
    tests/twoview.py's random scenes (outlier rates, thresholds, planar scenes, pure rotations ...),
  * the Hamming matcher (nearest neighbour, cross-check, knn + ratio) on random descriptor sets with many ties.
    python tests/scripts/soak_orb.py [--seconds 300] [--seed 1]
Prints progress lines and one JSON line; exits 1 at the first difference."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=300); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from visual_odometry_amd import _lib, geometry
    from visual_odometry_amd.detector import OrbDetector
    from visual_odometry_amd.matcher import HammingMatcher
    from oracle import oracle as O
    from test_gpu_orb import _fuzz_image
    from twoview import fuzz_problem
    ctx = _lib.default_context(0)
    ctx.set_poly_solver("opencv300")
    rng = np.random.default_rng(a.seed)
    t0 = tick = time.time(); n_det = n_geo = n_match = n_trunc = 0

    def fail(what, tag):
        print("MISMATCH", what, tag, "seed", a.seed, flush=True); sys.exit(1)

    while time.time() - t0 < a.seconds:
        if time.time() - tick > 30: tick = time.time(); print(f"... detector {n_det}, two-view {n_geo}, matcher {n_match} identical so far", flush=True)
        # ---- detector
        h, w = int(rng.integers(60, 420)), int(rng.integers(60, 560))
        nfeatures = int(rng.choice([30, 100, 500, 1500, 3000])); nlevels = int(rng.integers(1, 9))
        thr = int(rng.choice([5, 10, 20, 40])); score = int(rng.integers(0, 2)); sf = float(rng.choice([1.2, 1.2, 1.1, 1.35, 1.5]))
        img = _fuzz_image(rng, h, w)
        p = O.orb_params(nfeatures=nfeatures, nlevels=nlevels, fast_threshold=thr, score_type=score, scale_factor=sf)
        ref = O.orb_detect_and_compute(img, p)
        got = OrbDetector(nfeatures=nfeatures, nlevels=nlevels, fastThreshold=thr, scoreType=score, scaleFactor=sf).detect_arrays(img)
        tag = f"{h}x{w} nf {nfeatures} L {nlevels} t {thr} score {score} sf {sf}"
        if got["truncated"]:
            n_trunc += 1                                       # a capacity of DESIGN.md section 7 was exceeded: flagged (RuntimeWarning), the result is not claimed exact
        else:
            if ref["overflow"]: fail("oracle overflow, no flag", tag)
            for k in ("xy", "octave", "response", "angle", "size", "desc"):
                if not np.array_equal(got[k], ref[k]): fail("detector " + k, tag)
            n_det += 1
        # ---- matcher on the descriptors just made, against a shuffled / perturbed copy (ties: duplicated rows)
        d1 = ref["desc"]
        if len(d1) >= 8:
            d2 = d1[rng.permutation(len(d1))[: max(4, len(d1) * 3 // 4)]].copy()
            flip = rng.random(d2.shape) < 0.02
            d2 ^= (flip * rng.integers(0, 256, d2.shape)).astype(np.uint8)
            d2 = np.concatenate([d2, d2[: len(d2) // 5]])                      # exact duplicates: equal distances
            for cc, name in ((2, "crosscheck"), (0, "nearest")):
                rq, rt, rd = O.match_hamming(d1, d2, cc)
                ms = HammingMatcher(crossCheck=(cc == 2)).match(d1, d2)
                if not (np.array_equal([m.queryIdx for m in ms], rq) and np.array_equal([m.trainIdx for m in ms], rt) and np.array_equal([m.distance for m in ms], rd)):
                    fail("matcher " + name, tag)
            n_match += 1
        # ---- two-view geometry
        for _ in range(2):
            pr = fuzz_problem(rng)
            rc, Es, mask, ninl = O.find_essential_ransac(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], thresh=pr["thresh"])
            E, m = geometry.findEssentialMat(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], threshold=pr["thresh"])
            if rc != 0:
                if E is not None: fail("findEssentialMat verdict", pr["tag"])
                continue
            if not (np.array_equal(m.ravel(), mask) and np.array_equal(E, Es[0])): fail("findEssentialMat", pr["tag"])
            inl = mask > 0
            ng, Rr, tr, pm = O.recover_pose(Es[0], pr["p1"][inl], pr["p2"][inl], pr["K"])
            ng2, R2, t2, pm2 = geometry.recoverPose(E, pr["p1"][inl], pr["p2"][inl], pr["K"])
            if not (ng2 == ng and np.array_equal(R2, Rr) and np.array_equal(t2, tr) and np.array_equal(np.asarray(pm2).ravel() > 0, np.asarray(pm).ravel() > 0)):
                fail("recoverPose", pr["tag"])
            n_geo += 1
    ctx.set_poly_solver("fast")
    print(json.dumps({"detector_cases": n_det, "capacity_flagged": n_trunc, "matcher_cases": n_match, "two_view_cases": n_geo,
                      "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
