#!/usr/bin/env python3
"""The pair path as the reference runs it live (src/visual_slam.py:17-19,294-298): SIFT detectAndCompute per frame, BFMatcher
(NORM_L2, crossCheck=True), findEssentialMat, recoverPose, triangulatePoints — through the per-call ABI (host buffers in and
out of every call), timed per stage.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--frames", type=int, default=9)
    a = ap.parse_args()
    from visual_odometry_amd import _lib, synth, geometry
    from visual_odometry_amd.detector import SiftDetector
    from visual_odometry_amd.matcher import L2Matcher
    seq = synth.sequence(a.frames, a.width, a.height, cache_dir="/tmp")
    ctx = _lib.default_context(0)
    det = SiftDetector(ctx=ctx); bf = L2Matcher(crossCheck=True, ctx=ctx)
    K = seq["K"]
    feats = [det.detect_arrays(seq["frames"][0])]
    bf.match_arrays(feats[0]["desc"], feats[0]["desc"])
    t = {"sift": 0.0, "match_l2": 0.0, "essential": 0.0, "pose": 0.0, "triangulate": 0.0}
    nm = ni = 0
    for k in range(1, a.frames):
        t0 = time.perf_counter(); feats.append(det.detect_arrays(seq["frames"][k])); t["sift"] += time.perf_counter() - t0
        t0 = time.perf_counter(); qi, ti, dd = bf.match_arrays(feats[k - 1]["desc"], feats[k]["desc"]); t["match_l2"] += time.perf_counter() - t0
        p1 = feats[k - 1]["xy"][qi].astype(np.float64); p2 = feats[k]["xy"][ti].astype(np.float64)
        t0 = time.perf_counter(); E, mask = geometry.findEssentialMat(p1, p2, K, geometry.RANSAC, 0.99, 1.0, ctx=ctx); t["essential"] += time.perf_counter() - t0
        inl = mask.ravel() > 0
        t0 = time.perf_counter(); _, R, tt, _ = geometry.recoverPose(E, p1[inl], p2[inl], K, ctx=ctx); t["pose"] += time.perf_counter() - t0
        P0 = K @ np.eye(4)[:3]; P1 = K @ np.hstack([R, tt])
        t0 = time.perf_counter(); geometry.triangulatePoints(P1, P0, p1[inl].T, p2[inl].T, ctx=ctx); t["triangulate"] += time.perf_counter() - t0
        nm += len(qi); ni += int(inl.sum())
    n = a.frames - 1
    tot = sum(t.values())
    print(json.dumps({"workload": f"live pair path (SIFT + L2 cross-check + E-RANSAC + recoverPose + DLT), {a.width}x{a.height}, per-call ABI, {n} pairs",
                      "ms_per_pair": round(tot / n * 1e3, 2), "pairs_per_s": round(n / tot, 1),
                      "ms_per_stage": {k: round(v / n * 1e3, 2) for k, v in t.items()},
                      "keypoints_per_frame": int(np.mean([len(f["xy"]) for f in feats])), "matches_per_pair": nm // n, "inliers_per_pair": ni // n}))


if __name__ == "__main__":
    main()
