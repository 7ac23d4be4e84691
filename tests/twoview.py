"""Seeded two-view problems shared by the CPU and GPU geometry tests (not a test module)."""
import numpy as np


def rot(ax, ang):
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    k = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * k @ k


def scene(seed, n, noise=0.3, outliers=0.3):
    rng = np.random.default_rng(seed)
    K = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    R = rot(rng.normal(size=3), rng.uniform(0.01, 0.1))
    t = rng.normal(size=3); t /= np.linalg.norm(t)
    X = rng.uniform(-4, 4, (n, 3)) + np.array([0, 0, 10])
    p1 = (X / X[:, 2:]) @ K.T
    X2 = X @ R.T + t
    p2 = (X2 / X2[:, 2:]) @ K.T
    p1 = p1[:, :2] + rng.normal(0, noise, (n, 2)); p2 = p2[:, :2] + rng.normal(0, noise, (n, 2))
    out = rng.random(n) < outliers
    p2[out] += rng.uniform(-50, 50, (int(out.sum()), 2))
    return K, R, t, p1, p2


def fuzz_problem(rng):
    """One random two-view problem: random motion (incl. pure rotation and pure forward motion), planar or deep scene,
    8..1500 points, 0..85 % outliers, random intrinsics, threshold and confidence."""
    n = int(rng.choice([8, 20, 100, 400, 1500])); outl = float(rng.choice([0.0, 0.2, 0.5, 0.85]))
    f = float(rng.uniform(300, 1500))
    K = np.array([[f, 0, rng.uniform(200, 700)], [0, f * rng.uniform(0.9, 1.1), rng.uniform(150, 400)], [0, 0, 1]])
    ang = rng.normal(0, 0.15, 3); kx = np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]])
    R = np.eye(3) + kx + kx @ kx / 2
    u, _, vt = np.linalg.svd(R); R = u @ vt
    mode = int(rng.integers(0, 4))
    t = np.zeros(3) if mode == 0 else np.array([0, 0, 1.0]) if mode == 1 else rng.normal(size=3)
    X = rng.uniform(-3, 3, (n, 3)) + np.array([0, 0, 8.0])
    if mode == 3:
        X[:, 2] = 8.0                                               # fronto-parallel plane
    p1 = ((X / X[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.4, (n, 2))
    X2 = X @ R.T + 0.5 * t
    p2 = ((X2 / X2[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.4, (n, 2))
    bad = rng.random(n) < outl
    p2[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
    thresh = float(rng.choice([0.5, 1.0, 3.0])); prob = float(rng.choice([0.9, 0.99, 0.999]))
    return dict(K=K, p1=p1, p2=p2, thresh=thresh, prob=prob, tag=f"n {n} outl {outl} mode {mode}")


def five_point_sample(rng):
    R = rot(rng.normal(size=3), rng.uniform(0, 0.3)); t = rng.normal(size=3)
    X = rng.uniform(-2, 2, (5, 3)) + np.array([0, 0, 6])
    x1 = X[:, :2] / X[:, 2:]; X2 = X @ R.T + t; x2 = X2[:, :2] / X2[:, 2:]
    return x1, x2


def oracle_pair_stages(O, f1, f2, params, K, cross_check=2):
    """The oracle's stages composed in the order of visual_slam.py:294-298, keeping every intermediate the GPU path
    can be asked for (descriptors, match list, inlier mask), which voo_pair does not return."""
    d1, d2 = O.orb_detect_and_compute(f1, params), O.orb_detect_and_compute(f2, params)
    qi, ti, dist = O.match_hamming(d1["desc"], d2["desc"], cross_check)
    p1, p2 = d1["xy"][qi].astype(np.float64), d2["xy"][ti].astype(np.float64)
    rc, E, mask, ninl = O.find_essential_ransac(p1, p2, K)
    out = dict(d1=d1, d2=d2, qi=qi, ti=ti, dist=dist, rc=rc, mask=mask, n_inl=ninl)
    if rc != 0:
        return out
    inl = mask > 0
    ng, R, t, pm = O.recover_pose(E[0], p1[inl], p2[inl], K)
    P = K @ np.hstack([R.T, -R.T @ t]); P0 = K @ np.eye(3, 4)          # image_pair.py:319-323
    X = O.triangulate(P, P0, p1[inl].T, p2[inl].T)
    out.update(E=E[0], R=R, t=t, n_good=ng, X=X / X[3])
    return out
