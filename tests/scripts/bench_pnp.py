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

K = np.array([[802.832, 0, 565.427], [0, 802.832, 240.124], [0, 0, 1.0]])      # the reference's camera (test.g2o:1)


def problem(rng, n, outl):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = rng.uniform(0.05, 0.5)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
    t = np.array([0.3, -0.2, 30.0]) + rng.normal(0, 0.5, 3)
    X = np.concatenate([rng.uniform(-12, 12, (n, 2)), rng.uniform(-1.5, 1.5, (n, 1))], axis=1)   # ground with relief
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.5, (n, 2))
    bad = rng.random(n) < outl
    uv[bad] += rng.uniform(-100, 100, (int(bad.sum()), 2))
    return X, uv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--points", type=int, default=500)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--refine", choices=["cv2", "fast"], default="cv2", help="final pose: cv2's solvePnP(ITERATIVE) (default) or the fast minimiser")
    args = ap.parse_args()
    rng = np.random.default_rng(11)
    probs = [problem(rng, args.points, args.outliers) for _ in range(args.problems)]
    obj = np.concatenate([p[0] for p in probs]); img = np.concatenate([p[1] for p in probs])
    off = (np.arange(args.problems + 1) * args.points).astype(np.int32)
    ctx = _lib.default_context()
    ctx.set_pnp_refine(args.refine); O.set_pnp_refine(args.refine)
    geometry.solve_pnp_ransac_batch(obj, img, off, K, ctx=ctx)                   # warm-up
    ctx.check(ctx.lib.vo_profile_enable(ctx.handle, 1)); ctx.check(ctx.lib.vo_profile_reset(ctx.handle))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        status, rvec, tvec, mask, ninl = geometry.solve_pnp_ransac_batch(obj, img, off, K, ctx=ctx)
    dt = (time.perf_counter() - t0) / args.steps
    ms = np.zeros(_lib.VO_STAGE_COUNT, np.float32); cnt = np.zeros(_lib.VO_STAGE_COUNT, np.int32)
    ctx.check(ctx.lib.vo_profile_read(ctx.handle, ms.ctypes.data, cnt.ctypes.data))
    names = [ctx.lib.vo_stage_name(i).decode() for i in range(_lib.VO_STAGE_COUNT)]
    kernel_ms = float(ms[names.index("misc")] / max(cnt[names.index("misc")], 1))
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
