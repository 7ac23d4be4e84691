"""Checker script (GPU box; imports the oracle, so it lives under tests/): random PnP problems (sizes, outlier rates, noise, planar maps, near-planar maps) through the library and the
oracle in both refinement modes: identical inlier sets, and the largest pose difference seen."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import geometry, _lib
from oracle import oracle as O
K = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
ctx = _lib.default_context()
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
worst = {"cv2": 0.0, "fast": 0.0}; bad = 0; n_ok = 0; planar = 0
for it in range(int(os.environ.get("N", "300"))):
    n = int(rng.choice([6, 7, 9, 15, 40, 120, 500, 1500])); outl = float(rng.choice([0, 0.1, 0.3, 0.5])); noise = float(rng.choice([0, 0.3, 1.0]))
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = rng.uniform(0.05, 1.2)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
    t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4, 9)])
    X = rng.uniform(-2, 2, (n, 3))
    kind = it % 4
    if kind == 1: X[:, 2] = 0.3 * X[:, 0] - 0.2 * X[:, 1] + 1.0; planar += 1                      # exactly planar
    if kind == 2: X[:, 2] = 0.3 * X[:, 0] - 0.2 * X[:, 1] + 1.0 + rng.normal(0, 0.05, n)           # thin slab: on either side of the 1e-3 rule
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, noise, (n, 2)) if noise else ((Xc / Xc[:, 2:]) @ K.T)[:, :2]
    b = rng.random(n) < outl; uv[b] += rng.uniform(-100, 100, (int(b.sum()), 2))
    for mode in ("cv2", "fast"):
        O.set_pnp_refine(mode); ctx.set_pnp_refine(mode)
        rc, rv, tv, mask, ninl = O.solve_pnp_ransac(X, uv, K)
        ok, rvec, tvec, inl = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
        if ok != (rc == 0): bad += 1; print("verdict differs", it, mode); continue
        if not ok: continue
        n_ok += 1
        if not np.array_equal(inl.ravel(), np.nonzero(mask)[0]): bad += 1; print("inliers differ", it, mode, n, outl); continue
        d = max(np.abs(rvec.ravel() - rv).max(), np.abs(tvec.ravel() - tv).max())
        if not np.isfinite(d): d = 0.0 if (np.isnan(rv).any() == np.isnan(rvec).any()) else 1.0
        if d > 1e-6: print("pose differs", it, mode, n, outl, noise, kind, ninl, d)
        worst[mode] = max(worst[mode], d)
O.set_pnp_refine("cv2"); ctx.set_pnp_refine("cv2")
print("problems solved", n_ok, "planar", planar, "mismatches", bad, "worst pose difference", worst)
