"""CPU: properties that pin the oracle's stages independently of any golden vector."""
import numpy as np
import pytest

from conftest import random_image


def test_resize_identity_and_constant(oracle):
    img = random_image(1, 50, 70)
    assert np.array_equal(oracle.resize_linear_exact(img, 70, 50), img)          # scale 1: coefficients (256, 0)
    flat = np.full((60, 90), 137, np.uint8)
    assert np.all(oracle.resize_linear_exact(flat, 75, 50) == 137)               # weights sum to one exactly


def test_resize_is_separable_bilinear_within_rounding(oracle):
    img = random_image(2, 120, 160)
    out = oracle.resize_linear_exact(img, 133, 100).astype(np.float64)
    sy = 120 / 100; sx = 160 / 133
    ys = (np.arange(100) + 0.5) * sy - 0.5; xs = (np.arange(133) + 0.5) * sx - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, 118); x0 = np.clip(np.floor(xs).astype(int), 0, 158)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x0 + 1] * fx) * (1 - fy) + (f[y0 + 1][:, x0] * (1 - fx) + f[y0 + 1][:, x0 + 1] * fx) * fy
    assert np.abs(out - ref).max() <= 1.6        # 8-bit coefficient quantisation + final rounding


def test_fast_score_definition(oracle):
    """Score = the largest threshold at which the pixel is still a corner, NMS keeps strict local maxima."""
    img = random_image(3, 64, 80)
    s20 = oracle.fast_score_nms(img, 20)
    ys, xs = np.nonzero(s20)
    assert len(ys) > 0
    assert s20[:3].sum() == 0 and s20[-3:].sum() == 0 and s20[:, :3].sum() == 0 and s20[:, -3:].sum() == 0
    for y, x in list(zip(ys, xs))[:40]:
        s = int(s20[y, x])
        assert s >= 20
        nb = s20[y - 1:y + 2, x - 1:x + 2].astype(int).copy(); nb[1, 1] = 0
        assert nb.max() == 0                                  # two NMS survivors are never adjacent
        if s < 254:
            assert oracle.fast_score_nms(img, s)[y, x] in (s, 0)   # still a corner at threshold s (NMS may differ)
    # raising the threshold never creates corners
    s40 = oracle.fast_score_nms(img, 40)
    assert np.count_nonzero(s40) <= np.count_nonzero(s20)


def test_blur_constant_and_symmetry(oracle):
    flat = np.full((40, 50), 200, np.uint8)
    assert np.all(oracle.gaussian_blur7(flat) == 202)          # taps sum to 257: 200*257^2 >> 16 rounds to 202
    img = random_image(4, 45, 61)
    assert np.array_equal(oracle.gaussian_blur7(img[::-1, ::-1])[::-1, ::-1], oracle.gaussian_blur7(img))
    assert np.array_equal(oracle.gaussian_blur7(img.T.copy()).T, oracle.gaussian_blur7(img))


def test_detect_keeps_border_and_quota(oracle):
    img = random_image(5, 300, 400)
    p = oracle.orb_params(nfeatures=300)
    d = oracle.orb_detect_and_compute(img, p)
    lw, lh, ls, q = oracle.level_geometry(300, 400, p)
    assert len(d["xy"]) <= 300 + 32 and len(d["xy"]) > 100
    for l in range(8):
        m = d["octave"] == l
        if not m.any():
            continue
        x = d["xy"][m, 0] / ls[l]; y = d["xy"][m, 1] / ls[l]
        assert x.min() >= 30.99 and x.max() < lw[l] - 31 + 0.01 and y.min() >= 30.99 and y.max() < lh[l] - 31 + 0.01
        assert m.sum() >= min(q[l], m.sum())
        key = np.rint(y).astype(np.int64) * 100000 + np.rint(x).astype(np.int64)
        assert len(np.unique(key)) == len(key)                 # (cv2's own list order, the default: a permutation, no repeats)
    assert np.all(np.diff(d["octave"]) >= 0)
    assert np.all((d["angle"] >= 0) & (d["angle"] < 360.001))
    # the canonical order is the same SET, sorted by (y, x) inside a level
    oracle.set_keypoint_order("canonical")
    try:
        c = oracle.orb_detect_and_compute(img, p)
    finally:
        oracle.set_keypoint_order("cv2")
    assert len(c["xy"]) == len(d["xy"])
    for l in range(8):
        m = c["octave"] == l
        if m.any():
            key = np.rint(c["xy"][m, 1] / ls[l]).astype(np.int64) * 100000 + np.rint(c["xy"][m, 0] / ls[l]).astype(np.int64)
            assert np.all(np.diff(key) > 0)
    a = {tuple(r) for r in np.c_[d["xy"], d["octave"]].tolist()}; b = {tuple(r) for r in np.c_[c["xy"], c["octave"]].tolist()}
    assert a == b


def test_descriptor_rotation_covariance(oracle):
    """Rotating the image by 180 degrees rotates keypoint angles by 180 and keeps most descriptor bits."""
    img = random_image(6, 260, 260)
    p = oracle.orb_params(nfeatures=200, nlevels=1)
    a = oracle.orb_detect_and_compute(img, p)
    b = oracle.orb_detect_and_compute(img[::-1, ::-1].copy(), p)
    pos_b = {(259 - x, 259 - y): i for i, (x, y) in enumerate(b["xy"].astype(int).tolist())}
    hits = 0
    for i, (x, y) in enumerate(a["xy"].astype(int).tolist()):
        j = pos_b.get((x, y))
        if j is None:
            continue
        hits += 1
        da = (a["angle"][i] - b["angle"][j]) % 360
        assert abs(da - 180) < 1.0
        assert np.unpackbits(a["desc"][i] ^ b["desc"][j]).sum() <= 40
    assert hits > 50


def test_matcher_semantics(oracle):
    rng = np.random.default_rng(7)
    t = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    q = t[rng.permutation(60)[:40]].copy()
    qi, ti, d = oracle.match_hamming(q, t, 0)
    assert np.array_equal(qi, np.arange(40)) and np.all(d == 0) and np.all((t[ti] == q).all(axis=1))
    q1, t1, d1 = oracle.match_hamming(q, t, 1); q2, t2, d2 = oracle.match_hamming(q, t, 2)
    assert np.array_equal(q1, q2) and np.array_equal(t1, t2)   # distinct exact matches: both cross-check rules agree
    # the legacy rule (1) keeps a query as soon as SOME train row chose it; OpenCV 4.x's mutual rule (2) can drop it
    q = np.zeros((2, 32), np.uint8); t = np.zeros((2, 32), np.uint8)
    q[1, 0] = 0b1111; t[0, 0] = 0b1; t[1, 0] = 0xff; t[1, 1] = 0b1          # no ties: q1 -> t0 forward, t1 -> q1 backward
    a = oracle.match_hamming(q, t, 1); b = oracle.match_hamming(q, t, 2)
    assert a[0].tolist() == [0, 1] and a[1].tolist() == [0, 1] and b[0].tolist() == [0] and b[1].tolist() == [0]
    # ratio rule is strict
    q = np.zeros((1, 32), np.uint8); t = np.zeros((2, 32), np.uint8); t[0, 0] = 1; t[1, 0] = 3
    assert len(oracle.knn2_ratio_hamming(q, t, 0.5)[0]) == 0 and len(oracle.knn2_ratio_hamming(q, t, 0.51)[0]) == 1


def rot(ax, ang):
    ax = np.asarray(ax, float) / np.linalg.norm(ax)
    k = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * k + (1 - np.cos(ang)) * k @ k


def test_five_point_contains_true_essential(oracle):
    rng = np.random.default_rng(8)
    dists = []
    for _ in range(60):
        R = rot(rng.normal(size=3), rng.uniform(0, 0.3)); t = rng.normal(size=3); t /= np.linalg.norm(t)
        X = rng.uniform(-2, 2, (5, 3)) + np.array([0, 0, 6])
        x1 = X[:, :2] / X[:, 2:]; X2 = X @ R.T + t; x2 = X2[:, :2] / X2[:, 2:]
        Es = oracle.five_point(x1, x2)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Et = tx @ R; Et /= np.linalg.norm(Et)
        assert 1 <= len(Es) <= 10
        dists.append(min(min(np.linalg.norm(E - Et), np.linalg.norm(E + Et)) for E in Es))
        assert dists[-1] < 1e-3              # near-double roots of the degree-10 polynomial amplify rounding
        for E in Es:
            assert abs(np.linalg.norm(E) - 1) < 1e-12
            assert max(abs(np.r_[x2[i], 1] @ E @ np.r_[x1[i], 1]) for i in range(5)) < 1e-7
            assert np.linalg.norm(2 * E @ E.T @ E - np.trace(E @ E.T) * E) < 1e-2
    assert np.percentile(dists, 90) < 1e-8


def test_ransac_seed_and_iteration_rule(oracle):
    rng = np.random.default_rng(9)
    K = np.array([[700, 0, 300], [0, 700, 200], [0, 0, 1.0]])
    R = rot([0, 1, 0.2], 0.04); t = np.array([1, 0, 0.1]); t /= np.linalg.norm(t)
    X = rng.uniform(-3, 3, (300, 3)) + np.array([0, 0, 9])
    p1 = ((X / X[:, 2:]) @ K.T)[:, :2]; X2 = X @ R.T + t; p2 = ((X2 / X2[:, 2:]) @ K.T)[:, :2]
    rc, E, mask, n = oracle.find_essential_ransac(p1, p2, K)
    assert rc == 0 and n == 300 and mask.all()                     # noise free: the first sample explains everything
    a = oracle.find_essential_ransac(p1, p2, K); b = oracle.find_essential_ransac(p1, p2, K)
    assert np.array_equal(a[1], b[1])                              # fixed seed -> deterministic
    assert oracle.find_essential_ransac(p1[:4], p2[:4], K)[0] == -3    # < 5 points: cv2 returns None
    ng, Rr, tr, pm = oracle.recover_pose(E[0], p1, p2, K)
    assert ng == 300 and np.linalg.norm(Rr - R) < 1e-6 and np.linalg.norm(tr.ravel() - t) < 1e-6


def test_triangulation_recovers_points(oracle):
    rng = np.random.default_rng(10)
    K = np.array([[700, 0, 300], [0, 700, 200], [0, 0, 1.0]])
    R = rot([0.1, 1, 0], 0.1); t = np.array([[1.0], [0.2], [0.0]])
    X = rng.uniform(-3, 3, (50, 3)) + np.array([0, 0, 9])
    P0 = K @ np.eye(3, 4); P1 = K @ np.hstack([R, t])
    x0 = P0 @ np.vstack([X.T, np.ones(50)]); x1 = P1 @ np.vstack([X.T, np.ones(50)])
    Xr = oracle.triangulate(P0, P1, x0[:2] / x0[2], x1[:2] / x1[2])
    assert np.abs(Xr[:3] / Xr[3] - X.T).max() < 1e-8


@pytest.mark.parametrize("bad", ["first_level", "wta_k", "patch_size"])
def test_unsupported_params_are_rejected(oracle, bad):
    kw = {bad: {"first_level": 1, "wta_k": 3, "patch_size": 21}[bad]}
    with pytest.raises(RuntimeError):
        oracle.orb_detect_and_compute(random_image(11, 100, 100), oracle.orb_params(**kw))


def _introselect_py(a, nth):
    """libstdc++'s std::nth_element(first, nth, last, greater) on a list of (response, id), restated sequentially in
    Python from the algorithm's published description — a structurally independent check of the C++ oracle unit."""
    def gt(x, y): return x[0] > y[0]

    def adjust_heap(f0, hole, ln, v):
        top, sc = hole, hole
        while sc < (ln - 1) // 2:
            sc = 2 * (sc + 1)
            if gt(a[f0 + sc], a[f0 + sc - 1]): sc -= 1
            a[f0 + hole] = a[f0 + sc]; hole = sc
        if ln % 2 == 0 and sc == (ln - 2) // 2:
            sc = 2 * (sc + 1); a[f0 + hole] = a[f0 + sc - 1]; hole = sc - 1
        parent = (hole - 1) // 2
        while hole > top and gt(a[f0 + parent], v):
            a[f0 + hole] = a[f0 + parent]; hole = parent; parent = (hole - 1) // 2
        a[f0 + hole] = v

    first, last = 0, len(a)
    if first == last or nth == last: return
    depth = 2 * ((last - first).bit_length() - 1)
    while last - first > 3:
        if depth == 0:
            ln = nth + 1 - first
            if ln >= 2:
                parent = (ln - 2) // 2
                while True:
                    adjust_heap(first, parent, ln, a[first + parent])
                    if parent == 0: break
                    parent -= 1
            for i in range(nth + 1, last):
                if gt(a[i], a[first]):
                    v = a[i]; a[i] = a[first]; adjust_heap(first, 0, ln, v)
            a[first], a[nth] = a[nth], a[first]
            return
        depth -= 1
        A, B, C = first + 1, first + (last - first) // 2, last - 1
        if gt(a[A], a[B]): m = B if gt(a[B], a[C]) else C if gt(a[A], a[C]) else A
        else: m = A if gt(a[A], a[C]) else C if gt(a[B], a[C]) else B
        a[first], a[m] = a[m], a[first]
        f, l, p = first + 1, last, a[first]
        while True:
            while gt(a[f], p): f += 1
            l -= 1
            while gt(p, a[l]): l -= 1
            if not f < l: break
            a[f], a[l] = a[l], a[f]; f += 1
        if f <= nth: first = f
        else: last = f
    for i in range(first + 1, last):
        v = a[i]
        if gt(v, a[first]):
            a[first + 1:i + 1] = a[first:i]; a[first] = v
        else:
            nx = i - 1
            while gt(v, a[nx]): a[nx + 1] = a[nx]; nx -= 1
            a[nx + 1] = v


def test_cv2_order_unit_is_libstdcxx_retain_best(oracle):
    """oracle.retain_best_cv2 = cv::KeyPointsFilter::retainBest: nth_element, then partition of the tail by
    `response >= n-th response` (every tie kept); compared with the Python restatement above and with the defining
    properties of the result."""
    rng = np.random.default_rng(11)
    for it in range(400):
        n = int(rng.integers(1, 400)); n_points = int(rng.integers(0, n + 3))
        kind = it % 4
        r = (rng.integers(0, 12, n) if kind == 0 else np.arange(n) if kind == 1 else
             np.where(np.arange(n) < n // 2, np.arange(n), n - np.arange(n)) if kind == 2 else rng.normal(0, 1, n)).astype(np.float32)
        got = oracle.retain_best_cv2(r, n_points)
        if n <= n_points:
            assert got.tolist() == list(range(n)); continue            # untouched
        if n_points == 0:
            assert len(got) == 0; continue
        thr = np.sort(r)[::-1][n_points - 1]
        assert sorted(got.tolist()) == np.nonzero(r >= thr)[0].tolist()    # the kept SET: everything at or above the n-th response
        a = [(float(v), i) for i, v in enumerate(r)]
        _introselect_py(a, n_points - 1)
        amb = a[n_points - 1][0]
        f, l = n_points, n                                              # std::partition(begin + n, end, response >= amb)
        while True:
            while f != l and a[f][0] >= amb: f += 1
            if f == l: break
            l -= 1
            while f != l and not a[l][0] >= amb: l -= 1
            if f == l: break
            a[f], a[l] = a[l], a[f]; f += 1
        assert got.tolist() == [i for _, i in a[:f]], (it, n, n_points)


def test_sqrtf_is_strictly_increasing_on_integers_below_2_pow_22():
    """The matrix-core L2 matcher of SIFT rows (k_nn_l2i8) selects on the exact integer d^2, cv2's batchDistance on
    sqrtf(d^2) with strict `<`: the same selection iff no two integers in range share a float square root.  True below 2^22
    (and k_sb_descriptor flags any row whose norm would allow a larger d^2); false further up, which is why the bound exists."""
    n = np.arange(0, (1 << 22) + 1, dtype=np.float32)
    r = np.sqrt(n)
    assert r.dtype == np.float32 and np.all(np.diff(r) > 0)
    hi = np.sqrt(np.arange(1 << 23, (1 << 23) + 4096, dtype=np.float32))
    assert np.any(np.diff(hi) == 0)                                   # beyond the bound neighbouring integers do collide


def test_l2_distance_of_integer_rows_is_exact_in_any_order(oracle):
    """normL2Sqr_ on SIFT rows (integers 0..255 as float32): every partial sum is an integer below 2^24, so the float sum is the
    exact integer d^2 whatever the order — the identity the int8 matrix-core matcher rests on."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (40, 128)).astype(np.float32); b = rng.integers(0, 256, (50, 128)).astype(np.float32)
    qi, ti, d = oracle.match_l2(a, b, 0)
    d2 = ((a[:, None, :].astype(np.int64) - b[None, :, :].astype(np.int64)) ** 2).sum(2)
    assert np.array_equal(ti, d2.argmin(1)) and np.array_equal(d, np.sqrt(d2.min(1).astype(np.float32)))
    shifted = ((a[:, None, :].astype(np.int64) - 128) * (b[None, :, :].astype(np.int64) - 128)).sum(2)
    na = ((a.astype(np.int64) - 128) ** 2).sum(1); nb = ((b.astype(np.int64) - 128) ** 2).sum(1)
    assert np.array_equal(na[:, None] + nb[None, :] - 2 * shifted, d2)
