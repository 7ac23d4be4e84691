#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small seeded inputs with the CPU oracle's outputs.

PARITY UNPINNED: the reference (Samirez/Visual_odometry) ships no tests, fixtures or golden vectors for this
path and cv2 is not importable here, so these vectors come from oracle/libvoo.so itself.  They freeze the
oracle's behaviour (any later edit that changes a result is caught by tests/test_oracle_golden.py) and let the
GPU box check the HIP path against committed data without rebuilding anything.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    assert not O.get_dk_early_exit()                       # vectors come from the faithful 300-sweep root finder
    seq = synth.sequence(2, 320, 240)
    frames, K = seq["frames"], seq["K"]
    p = O.orb_params(nfeatures=300, nlevels=6)
    d = [O.orb_detect_and_compute(f, p) for f in frames]
    lv = O.pyramid(frames[0], p)
    qi, ti, md = O.match_hamming(d[0]["desc"], d[1]["desc"], 2)        # BFMatcher(crossCheck=True), OpenCV 4.x rule
    lq, lt, ld = O.match_hamming(d[0]["desc"], d[1]["desc"], 1)        # the legacy rule
    rq, rt, rd = O.knn2_ratio_hamming(d[0]["desc"], d[1]["desc"], 0.8)
    pr = O.pair(frames[0], frames[1], p, K)
    np.savez_compressed(
        os.path.join(OUT, "pair_320x240.npz"), frames=frames, K=K, nfeatures=300, nlevels=6,
        level1=lv[1], fast0=O.fast_score_nms(lv[0], 20), blur0=O.gaussian_blur7(lv[0]),
        xy0=d[0]["xy"], angle0=d[0]["angle"], response0=d[0]["response"], octave0=d[0]["octave"], desc0=d[0]["desc"],
        xy1=d[1]["xy"], desc1=d[1]["desc"],
        cc_q=qi, cc_t=ti, cc_d=md, legacy_q=lq, legacy_t=lt, legacy_d=ld, ratio_q=rq, ratio_t=rt, ratio_d=rd,
        n_match=pr["n_match"], n_inl=pr["n_inl"], n_good=pr["n_good"], R=pr["R"], t=pr["t"], E=pr["E"], X=pr["X"])
    # geometry-only vector: exact synthetic correspondences with outliers
    rng = np.random.default_rng(99)
    Kg = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    ax = np.array([0.2, 1.0, 0.1]); ax /= np.linalg.norm(ax)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(0.05) * kx + (1 - np.cos(0.05)) * kx @ kx
    t = np.array([1, 0.1, 0.05]); t /= np.linalg.norm(t)
    X = rng.uniform(-4, 4, (400, 3)) + np.array([0, 0, 10])
    p1 = ((X / X[:, 2:]) @ Kg.T)[:, :2] + rng.normal(0, 0.3, (400, 2))
    X2 = X @ R.T + t
    p2 = ((X2 / X2[:, 2:]) @ Kg.T)[:, :2] + rng.normal(0, 0.3, (400, 2))
    out = rng.random(400) < 0.3
    p2[out] += rng.uniform(-50, 50, (int(out.sum()), 2))
    rc, E, mask, ninl = O.find_essential_ransac(p1, p2, Kg)
    ng, Rr, tr, pm = O.recover_pose(E[0], p1[mask > 0], p2[mask > 0], Kg)
    np.savez_compressed(os.path.join(OUT, "geometry_400.npz"), K=Kg, p1=p1, p2=p2, R_true=R, t_true=t,
                        E=E[0], mask=mask, n_inl=ninl, n_good=ng, R=Rr, t=tr, pose_mask=pm)
    # frame ingest: cv2.resize(img, dim) INTER_LINEAR (visual_slam.py:346-352)
    rng = np.random.default_rng(123)
    big = rng.integers(0, 256, (108, 192, 3), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "ingest_192x108.npz"), src=big, dst_57x32=O.resize_linear(big, 57, 32),
                        dst_96x54=O.resize_linear(big, 96, 54), dst_250x120=O.resize_linear(big, 250, 120),
                        area_96x54=O.resize_area(big, 96, 54), area_64x36=O.resize_area(big, 64, 36), area_57x32=O.resize_area(big, 57, 32))
    # localisation: cv2.solvePnPRansac (visual_slam.py:231-235)
    rng = np.random.default_rng(77)
    Kp = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
    ax = np.array([0.3, -0.8, 0.5]); ax /= np.linalg.norm(ax)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rp = np.eye(3) + np.sin(0.35) * kx + (1 - np.cos(0.35)) * kx @ kx
    tp = np.array([0.2, -0.1, 5.5])
    Xp = rng.uniform(-2, 2, (240, 3)); Xc = Xp @ Rp.T + tp
    uvp = ((Xc / Xc[:, 2:]) @ Kp.T)[:, :2] + rng.normal(0, 0.4, (240, 2))
    badp = rng.random(240) < 0.35
    uvp[badp] += rng.uniform(-80, 80, (int(badp.sum()), 2))
    rc, rv, tv, mk, ni = O.solve_pnp_ransac(Xp, uvp, Kp)
    np.savez_compressed(os.path.join(OUT, "pnp_240.npz"), K=Kp, obj=Xp, img=uvp, R_true=Rp, t_true=tp, rc=rc, rvec=rv, tvec=tv,
                        mask=mk, n_inl=ni)
    # SIFT (the reference's live detector): a 96 x 128 crop of the synthetic frame, every keypoint and descriptor
    simg = np.ascontiguousarray(frames[0][60:156, 90:218])
    sr = O.sift_detect_and_compute(simg)
    np.savez_compressed(os.path.join(OUT, "sift_96x128.npz"), img=simg, n=sr["n_found"], xy=sr["xy"], size=sr["size"], angle=sr["angle"],
                        response=sr["response"], octave=sr["octave"], desc=sr["desc"].astype(np.uint8))
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
