"""CPU: the C oracle against independent numpy restatements of the same published definitions.

These do not pin the oracle to cv2 (parity stays unpinned, see oracle/voo.h); they catch slips of the C code by
computing every stage a second, structurally different way (brute force / textbook formulas / numpy SVD)."""
import os

import numpy as np
import pytest

from conftest import random_image

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _pattern():
    src = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "orb_pattern.inc")).read()
    body = src[src.index("*/") + 2:]
    return np.array([int(v) for v in body.replace("\n", " ").split(",") if v.strip()], np.int32).reshape(256, 4)


def test_gaussian_taps_from_the_formula(oracle):
    """getGaussianKernel(7, 2): exp(-x^2 / 8) normalised, scaled by 256 and rounded -> the integer taps."""
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    taps = np.rint(256 * g / g.sum()).astype(int)
    assert taps.tolist() == [18, 34, 49, 55, 49, 34, 18]
    img = random_image(1, 40, 52).astype(np.int64)
    pad = np.pad(img, 3, mode="reflect")                       # numpy 'reflect' == BORDER_REFLECT_101
    h = sum(taps[k] * pad[3:-3, k:k + 52] for k in range(7))
    hp = np.pad(h, ((3, 3), (0, 0)), mode="reflect")
    v = sum(taps[k] * hp[k:k + 40, :] for k in range(7))
    ref = np.minimum((v + (1 << 15)) >> 16, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian_blur7(img.astype(np.uint8)), ref)


def test_fast_score_brute_force(oracle):
    """cornerScore = the largest threshold t' for which the pixel is still a FAST-9 corner (checked arc by arc)."""
    img = random_image(2, 36, 44)
    f = img.astype(int)

    def is_corner(y, x, t):
        v = f[y, x]
        ring = [f[y + dy, x + dx] for dx, dy in RING]
        for s in range(16):
            arc = [ring[(s + j) % 16] for j in range(9)]
            if all(p > v + t for p in arc) or all(p < v - t for p in arc):
                return True
        return False

    raw = np.zeros_like(f)
    for y in range(3, 33):
        for x in range(3, 41):
            if is_corner(y, x, 20):
                t = 20
                while t < 255 and is_corner(y, x, t + 1):
                    t += 1
                raw[y, x] = t
    ref = np.zeros_like(f)
    for y in range(3, 33):
        for x in range(3, 41):
            s = raw[y, x]
            nb = raw[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = 0
            if s and s > nb.max():
                ref[y, x] = s
    assert np.array_equal(oracle.fast_score_nms(img, 20).astype(int), ref) and ref.max() > 0


def test_harris_angle_and_brief_by_the_textbook(oracle):
    img = random_image(3, 200, 240)
    p = oracle.orb_params(nfeatures=120, nlevels=1)
    d = oracle.orb_detect_and_compute(img, p)
    f = img.astype(np.int64)
    blur = oracle.gaussian_blur7(img).astype(np.int64)
    pat = _pattern()
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert len(d["xy"]) > 60
    for i in range(0, len(d["xy"]), 3):
        x, y = int(d["xy"][i, 0]), int(d["xy"][i, 1])
        # Harris: 7x7 block of 3x3 Sobel sums, float32 formula in OpenCV's order
        a = b = c = 0
        for yy in range(y - 3, y + 4):
            for xx in range(x - 3, x + 4):
                w = f[yy - 1:yy + 2, xx - 1:xx + 2]
                ix = int((w[:, 2] * [1, 2, 1]).sum() - (w[:, 0] * [1, 2, 1]).sum())
                iy = int((w[2, :] * [1, 2, 1]).sum() - (w[0, :] * [1, 2, 1]).sum())
                a += ix * ix; b += iy * iy; c += ix * iy
        f32 = np.float32
        scale = f32(1) / f32(4 * 7 * 255.0)
        s4 = scale * scale * scale * scale
        resp = (f32(a) * f32(b) - f32(c) * f32(c) - f32(0.04) * (f32(a) + f32(b)) * (f32(a) + f32(b))) * s4
        assert resp == d["response"][i]
        # intensity centroid over the radius-15 disc; fastAtan2 is within 0.3 degrees of atan2
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                m10 += u * f[y + v, x + u]; m01 += v * f[y + v, x + u]
        ang = np.degrees(np.arctan2(float(m01), float(m10))) % 360
        assert min(abs(ang - d["angle"][i]), 360 - abs(ang - d["angle"][i])) < 0.3
        # steered BRIEF with the oracle's own angle, float32 rotation, round half to even
        th = f32(d["angle"][i]) * f32(np.pi / 180.0)
        ca, sa = f32(np.cos(np.float64(th))), f32(np.sin(np.float64(th)))
        bits = []
        for (x1, y1, x2, y2) in pat:
            ax = int(np.rint(f32(x1) * ca - f32(y1) * sa)); ay = int(np.rint(f32(x1) * sa + f32(y1) * ca))
            bx = int(np.rint(f32(x2) * ca - f32(y2) * sa)); by = int(np.rint(f32(x2) * sa + f32(y2) * ca))
            bits.append(1 if blur[y + ay, x + ax] < blur[y + by, x + bx] else 0)
        ref = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert np.array_equal(ref, d["desc"][i])


def test_retain_best_set_semantics(oracle):
    """Per level the kept set is exactly {response >= quota-th largest} of the FAST-ranked candidates."""
    img = random_image(4, 260, 300)
    p = oracle.orb_params(nfeatures=80, nlevels=2)
    lw, lh, ls, quota = oracle.level_geometry(260, 300, p)
    d = oracle.orb_detect_and_compute(img, p)
    lv = oracle.pyramid(img, p)
    for l in range(2):
        score = oracle.fast_score_nms(lv[l], 20)[31:lh[l] - 31, 31:lw[l] - 31].astype(int)
        vals = np.sort(score[score > 0])[::-1]
        n_fast = 2 * quota[l]
        kept_fast = int((score >= vals[n_fast - 1]).sum()) if len(vals) > n_fast else len(vals)
        got = d["response"][d["octave"] == l]
        assert quota[l] <= len(got) <= kept_fast
        assert len(np.unique(d["xy"][d["octave"] == l], axis=0)) == len(got)


def test_matcher_against_numpy_popcount(oracle):
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8); t = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    t[:20] = q[rng.permutation(70)[:20]]
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    qi, ti, d = oracle.match_hamming(q, t, 0)
    assert np.array_equal(ti, D.argmin(axis=1)) and np.array_equal(d, D.min(axis=1))
    # OpenCV 4.x crossCheck: mutual nearest neighbours, first minimum in both directions
    fwd, rev = D.argmin(axis=1), D.argmin(axis=0)
    mutual = [i for i in range(70) if rev[fwd[i]] == i]
    qi, ti, d = oracle.match_hamming(q, t, 2)
    assert qi.tolist() == mutual and np.array_equal(ti, fwd[mutual]) and np.array_equal(d, D[mutual, fwd[mutual]])
    # legacy crossCheck: every train row votes for its nearest query; a query keeps the closest voter (lowest index)
    best = {}
    for j in range(90):
        qq = int(D[:, j].argmin()); dd = int(D[qq, j])
        if qq not in best or dd < best[qq][0]:
            best[qq] = (dd, j)
    qi, ti, d = oracle.match_hamming(q, t, 1)
    assert qi.tolist() == sorted(best) and ti.tolist() == [best[k][1] for k in sorted(best)]
    # ratio rule
    order = np.argsort(D, axis=1, kind="stable")
    keep = [i for i in range(70) if float(D[i, order[i, 0]]) < 0.8 * float(D[i, order[i, 1]])]
    qi, ti, d = oracle.knn2_ratio_hamming(q, t, 0.8)
    assert qi.tolist() == keep and np.array_equal(ti, order[keep, 0])


def _rot(ax, ang):
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    k = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * k @ k


def _scene(seed, n):
    rng = np.random.default_rng(seed)
    K = np.array([[800, 0, 320], [0, 800, 240], [0, 0, 1.0]])
    R = _rot(rng.normal(size=3), 0.06); t = rng.normal(size=3); t /= np.linalg.norm(t)
    X = rng.uniform(-4, 4, (n, 3)) + np.array([0, 0, 10])
    p1 = ((X / X[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.3, (n, 2))
    X2 = X @ R.T + t
    p2 = ((X2 / X2[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.3, (n, 2))
    out = rng.random(n) < 0.3
    p2[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
    return K, R, t, p1, p2


def test_ransac_control_flow_in_python(oracle):
    """RANSACPointSetRegistrator::run re-stated in Python around the oracle's five-point solver: OpenCV's MWC RNG,
    repeat rejection, float32 Sampson test, strict `>` and RANSACUpdateNumIters must give the very same mask."""
    K, R, t, p1, p2 = _scene(6, 300)
    n = len(p1)
    ifx, ify = 1 / K[0, 0], 1 / K[1, 1]
    x1 = np.stack([p1[:, 0] * ifx + (-K[0, 2] * ifx), p1[:, 1] * ify + (-K[1, 2] * ify)], 1)
    x2 = np.stack([p2[:, 0] * ifx + (-K[0, 2] * ifx), p2[:, 1] * ify + (-K[1, 2] * ify)], 1)
    thr = np.float32((1.0 / ((K[0, 0] + K[1, 1]) / 2)) ** 2)
    state = 0xFFFFFFFFFFFFFFFF

    def nxt():
        nonlocal state
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return state & 0xFFFFFFFF

    def sampson_mask(E):
        h1 = np.c_[x1, np.ones(n)]; h2 = np.c_[x2, np.ones(n)]
        Ex1 = h1 @ E.T; Etx2 = h2 @ E
        num = (h2 * Ex1).sum(1) ** 2
        err = (num / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2)).astype(np.float32)
        return err <= thr

    niters, best, best_mask, it = 1000, 0, None, 0
    while it < niters:
        idx = []
        while len(idx) < 5:
            v = nxt() % n
            if v not in idx:
                idx.append(v)
        for E in oracle.five_point(x1[idx], x2[idx]):
            m = sampson_mask(E); good = int(m.sum())
            if good > max(best, 4):
                best, best_mask = good, m
                ep = (n - good) / n
                denom = 1 - (1 - ep) ** 5
                if denom < np.finfo(float).tiny:
                    niters = 0
                else:
                    num, den = np.log(0.01), np.log(denom)
                    niters = niters if (den >= 0 or -num >= niters * -den) else int(np.rint(num / den))
        it += 1
    rc, E, mask, ninl = oracle.find_essential_ransac(p1, p2, K)
    assert rc == 0 and ninl == best and np.array_equal(mask.astype(bool), best_mask)


def test_recover_pose_and_triangulate_with_numpy_svd(oracle):
    K, R, t, p1, p2 = _scene(7, 250)
    rc, E, mask, _ = oracle.find_essential_ransac(p1, p2, K)
    q1, q2 = p1[mask > 0], p2[mask > 0]
    ng, Ro, to, pm = oracle.recover_pose(E[0], q1, q2, K)
    U, _, Vt = np.linalg.svd(E[0])
    if np.linalg.det(U) < 0: U = -U
    if np.linalg.det(Vt) < 0: Vt = -Vt
    W = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])
    cands = [(U @ W @ Vt, U[:, 2]), (U @ W.T @ Vt, U[:, 2]), (U @ W @ Vt, -U[:, 2]), (U @ W.T @ Vt, -U[:, 2])]
    n1 = (np.c_[q1, np.ones(len(q1))] @ np.linalg.inv(K).T)[:, :2]
    n2 = (np.c_[q2, np.ones(len(q2))] @ np.linalg.inv(K).T)[:, :2]

    def tri(P1, P2, a, b):
        A = np.stack([a[0] * P1[2] - P1[0], a[1] * P1[2] - P1[1], b[0] * P2[2] - P2[0], b[1] * P2[2] - P2[1]])
        return np.linalg.svd(A)[2][3]

    counts = []
    for Rc, tc in cands:
        P = np.c_[Rc, tc]; good = 0
        for a, b in zip(n1, n2):
            Q = tri(np.eye(3, 4), P, a, b)
            ok = Q[2] * Q[3] > 0
            Q = Q / Q[3]
            z2 = (P @ Q)[2]
            good += bool(ok and Q[2] < 50 and 0 < z2 < 50)
        counts.append(good)
    k = int(np.argmax(counts))
    assert ng == max(counts)
    assert np.linalg.norm(Ro - cands[k][0]) < 1e-9 and np.linalg.norm(to.ravel() - cands[k][1]) < 1e-9
    P1 = K @ np.c_[Ro.T, -Ro.T @ to]; P0 = K @ np.eye(3, 4)
    Xo = oracle.triangulate(P1, P0, q1.T, q2.T)
    for i in range(0, len(q1), 7):
        ref = tri(P1, P0, q1[i], q2[i])
        a, b = Xo[:, i] / np.linalg.norm(Xo[:, i]), ref / np.linalg.norm(ref)
        assert 1 - abs(a @ b) < 1e-10


# ------------------------------------------------------------------ "next" row: frame ingest (cv2.resize INTER_LINEAR)
def _bilinear_float(src, dw, dh):
    """Textbook half-pixel-centre bilinear in float64 (what INTER_LINEAR approximates in 11-bit fixed point)."""
    s = src.astype(np.float64)
    if s.ndim == 2:
        s = s[:, :, None]
    sh, sw = s.shape[:2]
    fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    xa, xb = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    ya, yb = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    ax = np.where((x0 < 0) | (x0 >= sw - 1), 0.0, ax)
    top = s[ya][:, xa] * (1 - ax)[None, :, None] + s[ya][:, xb] * ax[None, :, None]
    bot = s[yb][:, xa] * (1 - ax)[None, :, None] + s[yb][:, xb] * ax[None, :, None]
    out = top * (1 - ay)[:, None, None] + bot * ay[:, None, None]
    return out if src.ndim == 3 else out[:, :, 0]


@pytest.mark.parametrize("shape,dsize", [((216, 384, 3), (115, 64)), ((97, 131), (64, 48)), ((50, 70, 4), (140, 100)),
                                         ((60, 80, 3), (80, 60)), ((33, 47), (200, 9))])
def test_ingest_resize_close_to_float_bilinear(oracle, shape, dsize):
    rng = np.random.default_rng(31)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = oracle.resize_linear(img, dsize[0], dsize[1]).astype(np.float64)
    ref = _bilinear_float(img, dsize[0], dsize[1])
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1.0               # 11-bit weights, truncating shifts: within one grey level
    assert abs((got - ref).mean()) < 0.3                # and unbiased to first order


def test_ingest_resize_identities(oracle):
    rng = np.random.default_rng(32)
    img = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 52, 40), img)                 # same size: weights (2048, 0)
    half = oracle.resize_linear(img, 26, 20)                                      # exact 2:1 is the 2x2 mean (INTER_AREA)
    ref = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, ref.astype(np.uint8))
    flat = np.full((30, 30), 77, np.uint8)
    assert np.all(oracle.resize_linear(flat, 11, 17) == 77)                       # constants survive the fixed point


# ------------------------------------------------------------------ "next" row: PnP-RANSAC localisation (visual_slam.py:231-243)
def _pnp_problem(seed, n, outl, noise=0.5):
    rng = np.random.default_rng(seed)
    K = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = rng.uniform(0.1, 0.6)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
    t = np.array([0.3, -0.2, 6.0]) + rng.normal(0, 0.3, 3)
    X = rng.uniform(-2, 2, (n, 3))
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outl
    uv[bad] += rng.uniform(-100, 100, (int(bad.sum()), 2))
    return K, X, uv, R, t, bad


@pytest.mark.parametrize("seed,n,outl", [(1, 200, 0.3), (2, 50, 0.1), (3, 1000, 0.5), (4, 12, 0.0), (6, 5, 0.0)])
def test_pnp_ransac_recovers_the_pose_and_the_lm_minimum(oracle, seed, n, outl):
    """The restated solvePnPRansac finds the generating pose and inlier set; its refined pose is the reprojection-error
    minimum over those inliers, checked against scipy's own Levenberg-Marquardt in the rvec parametrisation cv2 uses."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    K, X, uv, R, t, bad = _pnp_problem(seed, n, outl)
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
    assert rc == 0 and ninl == int(mask.sum())
    assert ((mask > 0) == ~bad).mean() > 0.97
    tol = 0.02 if n > 6 else 0.2
    assert np.abs(oracle.rodrigues(rv) - R).max() < tol and np.abs(tv - t).max() < 5 * tol
    if n > 5:
        inl = mask > 0

        def res(p):
            Xc = X[inl] @ Rotation.from_rotvec(p[:3]).as_matrix().T + p[3:]
            return (((Xc / Xc[:, 2:]) @ K.T)[:, :2] - uv[inl]).ravel()
        sol = least_squares(res, np.concatenate([rv, tv]) + 1e-3, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert np.abs(sol.x[:3] - rv).max() < 1e-7 and np.abs(sol.x[3:] - tv).max() < 1e-7
    rc2, rv2, tv2, mask2, _ = oracle.solve_pnp_ransac(X, uv, K)             # fixed seed: deterministic
    assert np.array_equal(rv, rv2) and np.array_equal(tv, tv2) and np.array_equal(mask, mask2)


def test_pnp_final_pose_follows_cv2s_solvepnp_iterative(oracle):
    """solvePnPRansac's last step, solvePnP(inliers, SOLVEPNP_ITERATIVE) = cvFindExtrinsicCameraParams2 (oracle/voo_pnp.c
    pn_refine_cv2): the result is a stationary point of the pixel reprojection cost over the (float32) inliers; it equals
    the fast mode's minimum to 1e-6; a planar map takes the homography start and still lands on the generating pose; and with
    exactly 5 non-planar inliers cv2 keeps the RANSAC model ("DLT algorithm needs at least 6 points")."""
    def cost_grad(X, uv, K, rv, tv, mask):
        Xf = X.astype(np.float32).astype(np.float64)[mask > 0]; uf = uv.astype(np.float32).astype(np.float64)[mask > 0]
        def cost(p):
            Rm = oracle.rodrigues(p[:3]); Xc = Xf @ Rm.T + p[3:]
            pr = np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)
            return ((pr - uf) ** 2).sum()
        p = np.concatenate([rv, tv]); g = np.zeros(6)
        for k in range(6):
            d = np.zeros(6); d[k] = 1e-6
            g[k] = (cost(p + d) - cost(p - d)) / 2e-6
        return cost(p), g
    try:
        five = 0
        for seed in range(100, 170):
            n = [8, 20, 100, 400][seed % 4]; outl = [0, 0.2, 0.5][seed % 3]
            K, X, uv, R, t, bad = _pnp_problem(seed, n, outl)
            oracle.set_pnp_refine("cv2"); rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
            oracle.set_pnp_refine("fast"); rc0, rv0, tv0, mask0, _ = oracle.solve_pnp_ransac(X, uv, K)
            assert rc == rc0
            if rc:
                continue
            assert np.array_equal(mask, mask0)
            if ninl == 5:
                five += 1
                continue
            assert np.abs(rv - rv0).max() < 1e-6 and np.abs(tv - tv0).max() < 1e-6
            c0, g = cost_grad(X, uv, K, rv, tv, mask)
            assert np.abs(g).max() < 1e-3 * max(c0, 1.0)                # stationary (the cost is a sum of squared pixels)
        assert five >= 1
        oracle.set_pnp_refine("cv2")
        for seed in range(5):                                           # planar structure: homography start
            K, X, uv, R, t, bad = _pnp_problem(200 + seed, 150, 0.2)
            X[:, 2] = 0.25 * X[:, 0] + 0.1 * X[:, 1] + 0.5
            Xc = X @ R.T + t
            uv2 = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + np.random.default_rng(seed).normal(0, 0.3, uv.shape)
            uv2[bad] = uv[bad]
            rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv2, K)
            assert rc == 0 and ninl >= 100
            assert np.abs(oracle.rodrigues(rv) - R).max() < 0.01 and np.abs(tv - t).max() < 0.05
    finally:
        oracle.set_pnp_refine("cv2")


def test_pnp_error_codes_and_rodrigues(oracle):
    from scipy.spatial.transform import Rotation
    K, X, uv, *_ = _pnp_problem(3, 4, 0.0)
    assert oracle.solve_pnp_ransac(X[:3], uv[:3], K)[0] == -3                # cv2 asserts npoints >= 4
    assert oracle.solve_pnp_ransac(X, uv, K)[0] == 0                         # exactly 4: cv2's P3P branch (test_p3p_branch_...)
    rng = np.random.default_rng(8)
    for _ in range(50):
        r = rng.normal(0, 1.3, 3)
        Rm = oracle.rodrigues(r)
        assert np.abs(Rm - Rotation.from_rotvec(r).as_matrix()).max() < 1e-14
        assert np.abs(oracle.rodrigues(Rm) - Rotation.from_matrix(Rm).as_rotvec()).max() < 1e-12
    assert np.array_equal(oracle.rodrigues(np.zeros(3)), np.eye(3))
    near_pi = np.array([np.pi - 1e-9, 0, 0])                                  # the s < 1e-5 branch of cvRodrigues2
    assert np.abs(oracle.rodrigues(oracle.rodrigues(near_pi)) - near_pi).max() < 1e-6


def _area_weights(s, d):
    W = np.zeros((d, s)); sc = s / d
    for i in range(d):
        a, b = i * sc, (i + 1) * sc
        for j in range(int(np.floor(a)), min(int(np.ceil(b)), s)):
            W[i, j] = max(0.0, min(b, j + 1) - max(a, j))
        W[i] /= W[i].sum()
    return W


@pytest.mark.parametrize("dsize", [(192, 108), (128, 72), (96, 54), (96, 108), (115, 64), (384, 216), (383, 215), (1, 1)])
def test_inter_area_against_exact_area_mean(oracle, dsize):
    """cv2.resize(..., INTER_AREA) (image_and_keypoints.py:42) is the area-weighted mean of the covered source cells:
    the float32 restatement must round the float64 mean to the nearest grey level (ties aside)."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (216, 384, 3), dtype=np.uint8)
    out = oracle.resize_area(img, dsize[0], dsize[1]).astype(np.float64)
    Wx, Wy = _area_weights(384, dsize[0]), _area_weights(216, dsize[1])
    ref = np.einsum("yj,jxc->yxc", Wy, np.einsum("xi,jic->jxc", Wx, img.astype(np.float64)))
    assert np.abs(out - ref).max() <= 0.5 + 1e-4
    if dsize == (192, 108):                                   # 2 x 2 blocks round half up: (a + b + c + d + 2) >> 2
        blk = img.reshape(108, 2, 192, 2, 3).astype(np.int64).sum(axis=(1, 3))
        assert np.array_equal(out, (blk + 2) >> 2)
    with pytest.raises(NotImplementedError):
        oracle.resize_area(img, 500, 300)


def test_l2_matcher_against_numpy(oracle):
    """BFMatcher(NORM_L2) (visual_slam.py:19): float32 distances in normL2Sqr_'s order vs float64 numpy."""
    rng = np.random.default_rng(9)
    t = (rng.random((150, 128)) * 200).astype(np.float32)
    q = t[rng.permutation(150)[:100]] + rng.normal(0, 2, (100, 128)).astype(np.float32)
    D = np.sqrt(((q[:, None, :].astype(np.float64) - t[None].astype(np.float64)) ** 2).sum(2))
    qi, ti, d = oracle.match_l2(q, t, 0)
    assert np.array_equal(ti, D.argmin(axis=1)) and np.abs(d - D.min(axis=1)).max() < 1e-4
    fwd, rev = D.argmin(axis=1), D.argmin(axis=0)
    mutual = [i for i in range(100) if rev[fwd[i]] == i]
    qi, ti, d = oracle.match_l2(q, t, 2)
    assert qi.tolist() == mutual and np.array_equal(ti, fwd[mutual])


def test_p3p_branch_recovers_the_pose(oracle):
    """solvePnPRansac's four-point branch (P3P + fourth-point disambiguation): on exact projections the generating
    pose comes back, and the three solving points reproject to their pixels to float32 accuracy."""
    rng = np.random.default_rng(5)
    K = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
    good = 0
    for _ in range(100):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 1.0)
        kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx
        t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4, 8)])
        X = rng.uniform(-2, 2, (4, 3))
        Xc = X @ R.T + t
        uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2]
        rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
        assert rc == 0 and ninl == 4 and mask.tolist() == [1, 1, 1, 1]
        Rg = oracle.rodrigues(rv)
        Xg = X @ Rg.T + tv
        rep = ((Xg / Xg[:, 2:]) @ K.T)[:, :2]
        assert np.abs(rep[:3] - uv[:3]).max() < 1e-2             # the three points P3P solves with
        good += np.abs(Rg - R).max() < 1e-4 and np.abs(tv - t).max() < 1e-3
    assert good >= 95


def test_knn2_oracle_is_the_stable_two_smallest(oracle):
    """voo_knn2_hamming / voo_knn2_l2 (knnMatch(k=2), feature_detection.py:21) = the first two entries of a stable sort of every
    row of the distance matrix: batchDistance's insertion (ascending scan, strict <) stated another way."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    for q, t in ((rng.integers(0, 256, (90, 32), dtype=np.uint8), rng.integers(0, 256, (120, 32), dtype=np.uint8)),
                 (base[rng.integers(0, 10, 60)], base[rng.integers(0, 10, 40)])):
        D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
        o2 = np.argsort(D, axis=1, kind="stable")[:, :2]
        idx, dist = oracle.knn2_hamming(q, t)
        assert np.array_equal(idx, o2) and np.array_equal(dist, np.take_along_axis(D, o2, 1).astype(np.float32))
        for ratio in (0.3, 0.8, 1.0):                                    # and the ratio oracle is knn2 + the script's rule
            qi, ti, d = oracle.knn2_ratio_hamming(q, t, ratio)
            keep = dist[:, 0].astype(np.float64) < ratio * dist[:, 1].astype(np.float64)
            assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, idx[keep, 0]) and np.array_equal(d, dist[keep, 0])
    a = np.floor(rng.random((40, 128)) * 255).astype(np.float32); b = np.floor(rng.random((55, 128)) * 255).astype(np.float32)
    b[20:23] = b[0]; a[:2] = b[0]
    idx, dist = oracle.knn2_l2(a, b)
    D = np.sqrt(((a[:, None, :].astype(np.float64) - b[None, :, :]) ** 2).sum(2)).astype(np.float32)   # integer rows: the sums are exact
    o2 = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, o2) and np.array_equal(dist, np.take_along_axis(D, o2, 1))
    assert idx[0].tolist() == [0, 20] and dist[0].tolist() == [0.0, 0.0]
    qi, ti, d = oracle.match_l2(a, b, 0)
    assert np.array_equal(ti, idx[:, 0]) and np.array_equal(d, dist[:, 0])
    i1, d1 = oracle.knn2_l2(a, b[:1])
    assert np.all(i1[:, 1] == -1) and np.all(i1[:, 0] == 0)
