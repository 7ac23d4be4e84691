"""GPU parity of cv2's keypoint ORDER (vo_set_keypoint_order(ctx, 1)): the north star asks for bit-exact keypoint
INDICES and match pairs, and cv2's list order is the permutation cv::KeyPointsFilter::retainBest's std::nth_element +
std::partition leave behind (oracle/voo_cv2order.cpp calls those algorithms literally)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(seed, count):
    rng = np.random.default_rng(seed)
    for it in range(count):
        n = int(rng.integers(1, 30000 if it % 40 == 0 else 3000 if it % 5 == 0 else 300))
        kind = int(rng.integers(0, 7))
        rngsz = int(rng.choice([3, 20, 60, 255, 100000]))
        i = np.arange(n)
        if kind == 0: r = rng.integers(0, rngsz, n).astype(np.float32)               # FAST-like scores with ties
        elif kind == 1: r = i.astype(np.float32)                                     # ascending
        elif kind == 2: r = (n - i).astype(np.float32)                               # descending
        elif kind == 3: r = ((i * 7919) % rngsz).astype(np.float32)
        elif kind == 4: r = np.where(i < n // 2, i, n - i).astype(np.float32)        # organ pipe (drives introselect to its heap-select fallback)
        elif kind == 5: r = ((rng.integers(0, rngsz, n) - rngsz // 2) * 1e-7).astype(np.float32)   # Harris-like: tiny, signed
        else: r = rng.normal(0, 1e-5, n).astype(np.float32)                          # distinct floats
        yield r, int(rng.integers(0, n + 3))


def test_retain_best_permutation_equals_libstdcxx(oracle, ctx):
    """10^3 response lists (heavy ties, sorted runs, organ pipes, up to 30000 entries): the kept indices come out in
    exactly the order libstdc++ leaves them in."""
    bad = 0
    for r, n_points in _lists(7, 1000):
        want = oracle.retain_best_cv2(r, n_points)
        got = ctx.retain_best(r, n_points)
        ok = np.array_equal(got, want)
        bad += not ok
        assert ok, (len(r), n_points, got[:10], want[:10])
    assert bad == 0


def test_detect_and_compute_in_cv2_order(oracle, seq_small):
    from visual_odometry_amd.detector import OrbDetector
    img = seq_small["frames"][0]
    for nf, nl in ((500, 8), (1500, 5)):
        p = oracle.orb_params(nfeatures=nf, nlevels=nl)
        ref = oracle.orb_detect_and_compute(img, p)                         # cv2's order: the default of oracle and library
        oracle.set_keypoint_order("canonical")
        try:
            canon = oracle.orb_detect_and_compute(img, p)
        finally:
            oracle.set_keypoint_order("cv2")
        got = OrbDetector(nfeatures=nf, nlevels=nl).detect_arrays(img)
        assert not got["truncated"]
        for key in ("xy", "octave", "angle", "response", "size", "desc"):
            assert np.array_equal(got[key], ref[key]), key                 # same keypoints at the same INDICES
        assert not np.array_equal(ref["xy"], canon["xy"])                   # ... and that order is not the canonical one
        k1 = sorted(map(tuple, np.c_[canon["octave"], canon["xy"]].tolist()))
        k2 = sorted(map(tuple, np.c_[got["octave"], got["xy"]].tolist()))
        assert k1 == k2                                                     # same SET in both modes
        again = OrbDetector(nfeatures=nf, nlevels=nl, keypoint_order="canonical").detect_arrays(img)   # the other mode on the same context
        assert np.array_equal(again["xy"], canon["xy"]) and np.array_equal(again["desc"], canon["desc"])
        back = OrbDetector(nfeatures=nf, nlevels=nl).detect_arrays(img)    # ... and switching back restores cv2's
        assert np.array_equal(back["xy"], ref["xy"]) and np.array_equal(back["desc"], ref["desc"])


def test_pairs_in_cv2_order_have_cv2_match_indices(oracle):
    """Config-2 sized frames through the batched path in cv2 order: match index pairs, inlier mask and pose equal the
    oracle's run on its cv2-ordered lists."""
    from twoview import oracle_pair_stages
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(3, 1280, 720, cache_dir="/tmp")
    frames, K = seq["frames"], seq["K"]
    fe = FrontEnd(720, 1280, max_frames=3, max_pairs=2, nfeatures=2000, nlevels=8)                 # keypoint_order="cv2" is the default
    fe.upload(frames); fe.detect(0, 3)
    res, X = fe.run_pairs([[0, 1], [1, 2]], K, fe.make_opts(want_points=True))
    p = oracle.orb_params(nfeatures=2000, nlevels=8)
    oracle.set_dk_early_exit(True)
    try:
        ref = [oracle_pair_stages(oracle, frames[i], frames[i + 1], p, K) for i in range(2)]
    finally:
        oracle.set_dk_early_exit(False)
    for k, r in enumerate(ref):
        if k == 0:
            got = fe.features(0)
            assert not got["truncated"]
            assert np.array_equal(got["xy"], r["d1"]["xy"]) and np.array_equal(got["desc"], r["d1"]["desc"])
        qi, ti, dist, mask = fe.pair_matches(k)
        assert np.array_equal(qi, r["qi"]) and np.array_equal(ti, r["ti"]) and np.array_equal(dist, r["dist"]), k
        assert np.array_equal(mask, r["mask"]), k
        g = res[k]
        assert np.array_equal(g["E"].reshape(3, 3), r["E"]) and np.array_equal(g["R"].reshape(3, 3), r["R"]), k
