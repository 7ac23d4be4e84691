"""GPU parity: brute-force Hamming matcher (bit-exact index pairs) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descs(seed, n, dup_from=None, flip_bits=0):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if dup_from is not None:
        m = min(n, len(dup_from))
        d[:m] = dup_from[rng.permutation(len(dup_from))[:m]]
        for i in range(m):                      # perturb a few bits so distances are small but not all zero
            for b in rng.integers(0, 256, rng.integers(0, flip_bits + 1)):
                d[i, b // 8] ^= 1 << (b % 8)
    return d


def _matcher(**kw):
    from visual_odometry_amd.matcher import HammingMatcher
    return HammingMatcher(**kw)


@pytest.mark.parametrize("nq,nt,seed", [(500, 500, 1), (2000, 2000, 2), (1, 7, 3), (257, 1023, 4), (2200, 1900, 5)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_match_modes_bit_exact(oracle, ctx, nq, nt, seed, mode):
    t = _descs(seed, nt)
    q = _descs(seed + 100, nq, dup_from=t, flip_bits=40)
    m = _matcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1)
    gq, gt, gd = m.match_arrays(q, t)
    rq, rt, rd = oracle.match_hamming(q, t, mode)
    assert np.array_equal(gq, rq) and np.array_equal(gt, rt) and np.array_equal(gd, rd)


def test_ties_prefer_lowest_index(oracle, ctx):
    # many exact duplicates: every tie must resolve to the lowest index, in both directions
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, (16, 32), dtype=np.uint8)
    q = base[rng.integers(0, 16, 300)]
    t = base[rng.integers(0, 16, 280)]
    for mode in (0, 1, 2):
        m = _matcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1)
        gq, gt, gd = m.match_arrays(q, t)
        rq, rt, rd = oracle.match_hamming(q, t, mode)
        assert np.array_equal(gq, rq) and np.array_equal(gt, rt) and np.array_equal(gd, rd)


@pytest.mark.parametrize("ratio", [0.3, 0.75, 0.9, 1.0])
def test_knn2_ratio_bit_exact(oracle, ctx, ratio):
    t = _descs(11, 1500)
    q = _descs(12, 1700, dup_from=t, flip_bits=60)
    gq, gt, gd = _matcher().ratio_match_arrays(q, t, ratio)
    rq, rt, rd = oracle.knn2_ratio_hamming(q, t, ratio)
    assert np.array_equal(gq, rq) and np.array_equal(gt, rt) and np.array_equal(gd, rd)


def test_empty_and_single(oracle, ctx):
    m = _matcher(crossCheck=True)
    e = np.zeros((0, 32), np.uint8)
    one = _descs(1, 1)
    assert m.match(e, one) == [] and m.match(one, e) == []
    assert _matcher().ratio_match(one, one, 0.75) == []      # fewer than two train rows: no (m, n) pair
    got = m.match(one, one)
    assert len(got) == 1 and (got[0].queryIdx, got[0].trainIdx, got[0].distance) == (0, 0, 0.0)


def test_real_descriptors(oracle, ctx, seq_small):
    from visual_odometry_amd.detector import OrbDetector
    det = OrbDetector(nfeatures=500)
    d1 = det.detect_arrays(seq_small["frames"][0])["desc"]
    d2 = det.detect_arrays(seq_small["frames"][1])["desc"]
    gq, gt, gd = _matcher(crossCheck=True).match_arrays(d1, d2)
    rq, rt, rd = oracle.match_hamming(d1, d2, 2)
    assert len(gq) > 100
    assert np.array_equal(gq, rq) and np.array_equal(gt, rt) and np.array_equal(gd, rd)
    assert np.all(np.diff(gq) > 0)       # ascending queryIdx, as BFMatcher.match returns them


@pytest.mark.parametrize("nq,nt", [(15, 16), (16, 17), (17, 15), (63, 65), (64, 64), (65, 63), (511, 513), (512, 512),
                                   (513, 511), (1025, 31), (33, 1027), (2, 2), (3, 1)])
def test_sizes_around_the_mfma_tiles(oracle, ctx, nq, nt):
    """The matcher works on 16-row MFMA blocks, 64-row LDS stages and 512-row workgroups: every remainder case,
    all four selection modes, including the second-nearest bookkeeping of knn2."""
    t = _descs(nq * 7 + nt, nt)
    q = _descs(nq + nt * 3, nq, dup_from=t, flip_bits=30)
    for mode in (0, 1, 2):
        m = _matcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1)
        assert all(np.array_equal(a, b) for a, b in zip(m.match_arrays(q, t), oracle.match_hamming(q, t, mode)))
    for ratio in (0.6, 1.0):
        assert all(np.array_equal(a, b) for a, b in zip(_matcher().ratio_match_arrays(q, t, ratio),
                                                        oracle.knn2_ratio_hamming(q, t, ratio)))


def test_extreme_descriptors(oracle, ctx):
    """All-zero / all-one rows: distances 0 and 256 (the ends of the +1/-1 dot-product range)."""
    z = np.zeros((40, 32), np.uint8); o = np.full((37, 32), 255, np.uint8)
    mix = np.concatenate([z[:5], o[:5], _descs(3, 20)])
    for q, t in ((z, o), (o, z), (mix, o), (z, mix), (mix, mix[::-1].copy())):
        for mode in (0, 1, 2):
            m = _matcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1)
            assert all(np.array_equal(a, b) for a, b in zip(m.match_arrays(q, t), oracle.match_hamming(q, t, mode)))
        assert all(np.array_equal(a, b) for a, b in zip(_matcher().ratio_match_arrays(q, t, 0.9),
                                                        oracle.knn2_ratio_hamming(q, t, 0.9)))


@pytest.mark.parametrize("kernel", ["popcount", "mfma"])
def test_popcount_kernel_gives_the_same_matches(oracle, kernel):
    """The XOR + popcount kernel (north_star's formulation), the int8 matrix-core kernel and the default FP4 one are
    interchangeable."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.matcher import HammingMatcher
    c = _lib.Context(0)
    try:
        c.set_matcher_kernel(kernel)
        for nq, nt in ((2000, 2000), (513, 31), (17, 900), (1, 1)):
            t = _descs(nq + 5, nt)
            q = _descs(nt + 9, nq, dup_from=t, flip_bits=40)
            for mode in (0, 1, 2):
                m = HammingMatcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1, ctx=c)
                assert all(np.array_equal(a, b) for a, b in zip(m.match_arrays(q, t), oracle.match_hamming(q, t, mode)))
            m = HammingMatcher(ctx=c)
            assert all(np.array_equal(a, b) for a, b in zip(m.ratio_match_arrays(q, t, 0.8), oracle.knn2_ratio_hamming(q, t, 0.8)))
        with pytest.raises(Exception):
            c.check(c.lib.vo_set_matcher_kernel(c.handle, 7))
    finally:
        c.close()


def test_fp4_matrix_core_kernel_gives_the_same_matches(oracle, seq_small):
    """The block-scaled FP4 form (v_mfma_scale_f32_16x16x128_f8f6f4, index carried in the FP32 accumulator) against the oracle:
    all modes, tile / stage / workgroup remainders, exact ties, and whole pairs through the batched path."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd
    from visual_odometry_amd.matcher import HammingMatcher
    c = _lib.Context(0)
    try:
        c.set_matcher_kernel("mfma_fp4")
        for nq, nt in ((2000, 2000), (2256, 2255), (513, 31), (17, 900), (1, 1), (64, 65), (1000, 15), (16, 16)):
            t = _descs(nq + 5, nt)
            q = _descs(nt + 9, nq, dup_from=t, flip_bits=40)
            for mode in (0, 1, 2):
                m = HammingMatcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1, ctx=c)
                assert all(np.array_equal(a, b) for a, b in zip(m.match_arrays(q, t), oracle.match_hamming(q, t, mode))), (nq, nt, mode)
            m = HammingMatcher(ctx=c)
            assert all(np.array_equal(a, b) for a, b in zip(m.ratio_match_arrays(q, t, 0.8), oracle.knn2_ratio_hamming(q, t, 0.8)))
        z = np.zeros((40, 32), np.uint8); o = np.full((33, 32), 255, np.uint8)           # distances 0 and 256, every row a tie
        for q, t in ((z, z), (z, o), (o, z)):
            m = HammingMatcher(crossCheck=True, ctx=c)
            assert all(np.array_equal(a, b) for a, b in zip(m.match_arrays(q, t), oracle.match_hamming(q, t, 2)))
        frames, K = seq_small["frames"], seq_small["K"]
        fe = FrontEnd(480, 640, max_frames=3, max_pairs=2, nfeatures=500, ctx=c)
        fe.upload(frames[:3]); fe.detect(0, 3)
        res = fe.run_pairs([[0, 1], [1, 2]], K)[0]
        p = oracle.orb_params(nfeatures=500)
        for i in range(2):
            r = oracle.pair(frames[i], frames[i + 1], p, K, want_points=False)
            assert (int(res["n_match"][i]), int(res["n_inl"][i])) == (r["n_match"], r["n_inl"])
            assert np.allclose(res["R"][i].reshape(3, 3), r["R"], rtol=0, atol=1e-6)   # (the default root finder vs the oracle's 300 sweeps)
    finally:
        c.close()


@pytest.mark.parametrize("dim,nq,nt", [(128, 700, 900), (128, 65, 64), (64, 300, 257), (61, 130, 70), (16, 40, 40), (3, 10, 200)])
def test_l2_matcher_bit_exact(oracle, ctx, dim, nq, nt):
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck) on float rows (visual_slam.py:19, the reference's live matcher): indices AND
    float distances equal the oracle's in every cross-check mode, including exact duplicates (ties go to the lowest index)
    and dimensions with a scalar tail (61) or no 16-element block at all (3)."""
    from visual_odometry_amd.matcher import BFMatcher, L2Matcher, NORM_L2
    rng = np.random.default_rng(dim * 1000 + nq)
    t = (rng.random((nt, dim), dtype=np.float32) * 255).astype(np.float32)
    q = t[rng.integers(0, nt, nq)] + rng.normal(0, 3, (nq, dim)).astype(np.float32)
    q[: nq // 8] = t[: nq // 8]                              # exact copies: distance 0
    t[nt // 2:nt // 2 + 5] = t[0]                            # duplicated train rows: ties
    for mode in (0, 1, 2):
        m = L2Matcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1, ctx=ctx)
        gq, gt, gd = m.match_arrays(q, t)
        rq, rt, rd = oracle.match_l2(q, t, mode)
        assert np.array_equal(gq, rq) and np.array_equal(gt, rt) and np.array_equal(gd, rd), mode
    ms = BFMatcher(NORM_L2, crossCheck=True).match(q, t)
    assert ms[0].distance == 0.0 and ms[0].trainIdx == 0 and all(a.queryIdx < b.queryIdx for a, b in zip(ms, ms[1:]))
    e = np.zeros((0, dim), np.float32)
    assert L2Matcher(ctx=ctx).match(e, t) == [] and L2Matcher(ctx=ctx).match(q, e) == []


def test_matcher_kernel_switch_after_detection_stays_consistent(oracle, seq_small):
    """The detector writes the operand image of the matrix-core kernel selected at that time; switching the kernel
    afterwards must not make the matcher read that image in the other format."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd
    frames, K = seq_small["frames"], seq_small["K"]
    c = _lib.Context(0)
    try:
        fe = FrontEnd(480, 640, max_frames=2, max_pairs=1, nfeatures=500, ctx=c)
        p = oracle.orb_params(nfeatures=500)
        d = [oracle.orb_detect_and_compute(frames[i], p)["desc"] for i in range(2)]
        want = oracle.match_hamming(d[0], d[1], 2)
        for first, then in (("mfma_fp4", "mfma"), ("mfma", "mfma_fp4"), ("mfma_fp4", "popcount"), ("popcount", "mfma_fp4")):
            c.set_matcher_kernel(first)
            fe.upload(frames[:2]); fe.detect(0, 2)
            c.set_matcher_kernel(then)
            fe.run_pairs([[0, 1]], K)
            qi, ti, dd, _ = fe.pair_matches(0)
            assert np.array_equal(qi, want[0]) and np.array_equal(ti, want[1]) and np.array_equal(dd, want[2]), (first, then)
    finally:
        c.close()


def _script_ratio_loop(matches, ratio):
    """/root/reference/src/feature_detection.py:21-26 (and :90-96), re-typed: the loop the script runs on knnMatch's result."""
    good = []
    for m, n in matches:
        if m.distance < ratio * n.distance:
            good.append(m)
    return good


@pytest.mark.parametrize("kernel", ["mfma_fp4", "mfma", "popcount"])
def test_knn_match_k2_hamming(oracle, kernel):
    """matcher.knnMatch(d1, d2, k=2) returns [DMatch, DMatch] rows — both neighbours, index and distance, as batchDistance's
    K = 2 insertion orders them (ties keep their train order) — in every Hamming kernel; the script's own loop over that
    result gives ratio_match's list."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.matcher import HammingMatcher
    c = _lib.Context(0)
    try:
        c.set_matcher_kernel(kernel)
        m = HammingMatcher(ctx=c)
        rng = np.random.default_rng(11)
        base = rng.integers(0, 256, (12, 32), dtype=np.uint8)
        cases = [(_descs(3, 700, dup_from=_descs(4, 900), flip_bits=40), _descs(4, 900)),
                 (base[rng.integers(0, 12, 200)], base[rng.integers(0, 12, 150)]),         # exact duplicates: ties in both entries
                 (_descs(5, 2100), _descs(6, 2300)), (_descs(7, 33), _descs(8, 2)), (_descs(9, 1), _descs(10, 300))]
        for q, t in cases:
            gi, gd = m.knn2_arrays(q, t)
            oi, od = oracle.knn2_hamming(q, t)
            assert np.array_equal(gi, oi) and np.array_equal(gd, od), (kernel, len(q), len(t))
            D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2) if len(q) * len(t) <= 200 * 150 else None
            if D is not None:                                                              # an independent statement of the rule
                o2 = np.argsort(D, axis=1, kind="stable")[:, :2]
                assert np.array_equal(gi, o2) and np.array_equal(gd, np.take_along_axis(D, o2, 1).astype(np.float32))
            rows = m.knnMatch(q, t, k=2)
            assert len(rows) == len(q) and all(len(r) == 2 for r in rows)
            assert all(r[0].queryIdx == i and r[1].queryIdx == i for i, r in enumerate(rows))
            for ratio in (0.3, 0.8):                                                       # the script's literal is 0.3
                good = _script_ratio_loop(rows, ratio)
                want = m.ratio_match(q, t, ratio)
                assert [(g.queryIdx, g.trainIdx, g.distance) for g in good] == [(w.queryIdx, w.trainIdx, w.distance) for w in want]
        # one train row: cv2 returns one-entry lists (and the script's `for m, n in` would raise on them)
        rows = m.knnMatch(_descs(1, 5), _descs(2, 1), k=2)
        assert [len(r) for r in rows] == [1] * 5 and all(r[0].trainIdx == 0 for r in rows)
        assert m.knnMatch(_descs(1, 5), np.zeros((0, 32), np.uint8), k=2) == [[]] * 5
        assert [len(r) for r in m.knnMatch(_descs(1, 5), _descs(2, 9), k=1)] == [1] * 5
        with pytest.raises(ValueError):
            HammingMatcher(crossCheck=True, ctx=c).knnMatch(_descs(1, 5), _descs(2, 9), k=2)
    finally:
        c.close()


@pytest.mark.parametrize("dim,nq,nt", [(128, 700, 900), (128, 2100, 2300), (128, 65, 64), (61, 130, 70), (3, 10, 200), (128, 40, 1)])
def test_knn_match_k2_l2(oracle, ctx, dim, nq, nt):
    """cv2.BFMatcher(cv2.NORM_L2).knnMatch(d1, d2, k=2) on float rows — what src/feature_detection.py:7-8,21 runs on SIFT
    descriptors: both neighbours and their float distances equal the oracle's, ties included; the script's loop over the
    rows = L2Matcher.ratio_match (vo_knn2_ratio_l2)."""
    from visual_odometry_amd.matcher import L2Matcher
    rng = np.random.default_rng(dim * 77 + nq)
    t = np.floor(rng.random((nt, dim), dtype=np.float32) * 255).astype(np.float32)       # SIFT rows hold integers 0..255
    q = np.clip(t[rng.integers(0, nt, nq)] + np.rint(rng.normal(0, 6, (nq, dim))), 0, 255).astype(np.float32)
    if nt > 8:
        t[nt // 2:nt // 2 + 4] = t[0]                                                      # duplicated train rows: ties
        q[:3] = t[0]
    m = L2Matcher(ctx=ctx)
    gi, gd = m.knn2_arrays(q, t)
    oi, od = oracle.knn2_l2(q, t)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    rows = m.knnMatch(q, t, k=2)
    assert len(rows) == nq and all(len(r) == min(2, nt) for r in rows)
    if nt >= 2:
        assert rows[0][0].trainIdx == 0 or nt <= 8
        for ratio in (0.3, 0.8):
            good = _script_ratio_loop(rows, ratio)
            want = m.ratio_match(q, t, ratio)
            assert [(g.queryIdx, g.trainIdx, g.distance) for g in good] == [(w.queryIdx, w.trainIdx, w.distance) for w in want]
    else:
        assert m.ratio_match(q, t, 0.8) == []
