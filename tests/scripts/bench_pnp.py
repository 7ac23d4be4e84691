#!/usr/bin/env python3
"""Measurement of the localisation row (cv2.solvePnPRansac, src/visual_slam.py:231-235): B independent problems per
launch on one MI355X, beside the CPU oracle (scalar C restatement) on a bounded sample.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O  # noqa: E402
from visual_odometry_amd import _lib, geometry  # noqa: E402

from tools import bench_passes as BP  # noqa: E402

K = BP.PNP_K                                                                    # the reference's camera (test.g2o:1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--points", type=int, default=500)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--refine", choices=["cv2", "fast"], default="cv2", help="final pose: cv2's solvePnP(ITERATIVE) (default) or the fast minimiser")
    args = ap.parse_args()
    ctx = _lib.default_context()
    ctx.set_pnp_refine(args.refine); O.set_pnp_refine(args.refine)
    line, raw = BP.pnp_pass(ctx, args.problems, args.points, args.outliers, args.steps)        # the pass bench.py's config.pnp runs
    probs, off, status, rvec, tvec, mask, ninl = (raw[k] for k in ("probs", "off", "status", "rvec", "tvec", "mask", "ninl"))
    dt, kernel_ms = line["ms_per_launch_with_copies"] / 1e3, line["kernel_ms_per_launch"]
    n_cpu = min(args.problems, 24)
    t1 = time.perf_counter()
    same = 0
    worst = 0.0
    for b in range(n_cpu):
        rc, rv, tv, m, ni = O.solve_pnp_ransac(probs[b][0], probs[b][1], K)
        same += int(rc == status[b] and np.array_equal(m, mask[off[b]:off[b + 1]]))
        if rc == 0:
            worst = max(worst, float(np.abs(rv - rvec[b]).max()), float(np.abs(tv - tvec[b]).max()))
    cpu = (time.perf_counter() - t1) / n_cpu
    print(json.dumps({
        "metric": "solvePnPRansac problems/s", "value": round(args.problems / dt, 1), "unit": "problems/s", "n_gpus": 1,
        "config": {"problems_per_launch": args.problems, "points_per_problem": args.points, "outlier_fraction": args.outliers,
                   "iterations": 100, "reprojection_error_px": 8.0, "refine": args.refine},
        "ms_per_launch_with_copies": round(1000 * dt, 3), "kernel_ms_per_launch": round(kernel_ms, 3),
        "ok_fraction": float((status == 0).mean()), "mean_inliers": float(ninl.mean()),
        "cpu_baseline": {"value": round(1.0 / cpu, 1), "unit": "problems/s", "cores": 1, "kind": "port",
                         "sample": f"{n_cpu} of the same problems through oracle/libvoo.so, one thread",
                         "identical_inlier_sets": f"{same}/{n_cpu}", "max_abs_pose_difference": worst}}))


if __name__ == "__main__":
    main()
