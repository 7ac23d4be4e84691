"""The reference's LIVE configuration as the batched, HBM-resident path (FrontEnd(detector="sift")): cv2.SIFT_create()
(/root/reference/src/visual_slam.py:17) + cv2.BFMatcher(cv2.NORM_L2, crossCheck=True) (:19) + the pair geometry of
:294-298, frame-batched on the MI355X, against the CPU oracle (oracle/voo_sift.c, voo_match.c, voo_geom.c).

 * keypoints (position, size, angle, response, packed octave) and descriptors of every frame of a batch: bit-identical to
   the oracle's per-image SIFT at the reference's working size (1152x648, visual_slam.py:34,346-352) and at 1280x720;
 * batched = per-image call (vo_sift_detect_and_compute, which runs the same kernels with one frame) = oracle;
 * the L2 cross-check matches the int8 matrix-core matcher selects = oracle.match_l2 on the oracle's float descriptors
   (indices AND float distances), also in nearest-neighbour, legacy-cross-check and ratio mode;
 * E, inlier mask, R, t of every pair = the oracle's findEssentialMat / recoverPose on the same correspondences."""
import numpy as np
import pytest

from conftest import random_image

pytestmark = pytest.mark.gpu

KEYS = ("xy", "size", "angle", "response", "octave")


def _flight(n, w, h):
    from visual_odometry_amd import synth
    return synth.sequence(n, w, h, cache_dir="/tmp")


def _same_features(got, want):
    assert len(got["xy"]) == want["n_found"] and want["n_found"] > 0
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert got["desc"].dtype == np.float32 and np.array_equal(got["desc"], want["desc"])


@pytest.mark.parametrize("w,h,n", [(1152, 648, 4), (1280, 720, 4)])
def test_batched_sift_equals_oracle_at_working_size(oracle, w, h, n):
    from visual_odometry_amd.frontend import FrontEnd
    seq = _flight(n, w, h)
    fe = FrontEnd(h, w, max_frames=n, max_pairs=n - 1, detector="sift")
    fe.upload(seq["frames"])
    fe.detect(0, n)
    feats = [fe.features(s) for s in range(n)]
    for s in range(n):
        assert not feats[s]["truncated"]
        _same_features(feats[s], oracle.sift_detect_and_compute(seq["frames"][s]))
    # the per-image call runs the same kernels with one frame
    from visual_odometry_amd.detector import SiftDetector
    one = SiftDetector(ctx=fe.ctx).detect_arrays(seq["frames"][1])
    for k in KEYS + ("desc",):
        assert np.array_equal(one[k], feats[1][k]), k


def _check_pairs(oracle, fe, feats, pairs, K, match_mode, oracle_mode, ratio=0.75):
    from visual_odometry_amd import frontend as F
    opts = fe.make_opts(match_mode=match_mode, ratio=ratio, want_points=True)
    res, X = fe.run_pairs(pairs, K, opts)
    res = res.copy(); X = X.copy()
    for p, (a, b) in enumerate(pairs):
        qi, ti, dd, mask = fe.pair_matches(p)
        if oracle_mode == "ratio":
            # knnMatch(k=2) + `m.distance < ratio * n.distance` on the float distances (src/feature_detection.py:20-26)
            d = np.sqrt(((feats[a]["desc"][:, None, :].astype(np.float64) - feats[b]["desc"][None, :, :]) ** 2).sum(2)).astype(np.float32) \
                if len(feats[a]["desc"]) * len(feats[b]["desc"]) < 4_000_000 else None
            if d is None:
                continue
            order = np.argsort(d, axis=1, kind="stable")[:, :2]
            d0 = d[np.arange(len(d)), order[:, 0]]; d1 = d[np.arange(len(d)), order[:, 1]]
            keep = d0.astype(np.float64) < ratio * d1.astype(np.float64)
            assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, order[keep, 0]) and np.array_equal(dd, d0[keep])
            continue
        oq, ot, od = oracle.match_l2(feats[a]["desc"], feats[b]["desc"], oracle_mode)
        assert res["n_match"][p] == len(oq)
        assert np.array_equal(qi, oq) and np.array_equal(ti, ot), (p, match_mode)
        assert np.array_equal(dd, od)                                   # the float distances cv2's DMatch carries
        p1 = feats[a]["xy"][qi].astype(np.float64); p2 = feats[b]["xy"][ti].astype(np.float64)
        rc, Es, omask, ninl = oracle.find_essential_ransac(p1, p2, K)
        assert rc == 0 and res["status"][p] == 0
        assert res["n_inl"][p] == ninl and np.array_equal(mask, omask)
        inl = omask > 0
        ng, R, t, pm = oracle.recover_pose(Es[0], p1[inl], p2[inl], K)
        assert np.array_equal(res["E"][p].reshape(3, 3), Es[0])
        assert res["n_good"][p] == ng and np.array_equal(res["R"][p].reshape(3, 3), R) and np.array_equal(res["t"][p].reshape(3, 1), t)
        P1 = K @ np.hstack([R.T, -R.T @ t]); P0 = K @ np.eye(3, 4)      # image_pair.py:319-323
        Xo = oracle.triangulate(P1, P0, p1[inl].T, p2[inl].T)
        Xo = Xo / Xo[3]
        nrm = np.maximum(np.linalg.norm(Xo[:3], axis=0), 1e-12)
        rel = np.linalg.norm(X[p][:3, :ninl] - Xo[:3], axis=0) / nrm
        assert rel[nrm <= 10 * np.median(nrm)].max() < 1e-6
    return res


def test_live_pair_path_batched_identical_to_oracle(oracle, kernel_dk_rule):
    """SIFT + L2 cross-check + E-RANSAC + recoverPose + DLT of consecutive 1152x648 frames, every pair compared."""
    from visual_odometry_amd import frontend as F
    w, h, n = 1152, 648, 4
    seq = _flight(n, w, h)
    fe = F.FrontEnd(h, w, max_frames=n, max_pairs=n, detector="sift")
    fe.upload(seq["frames"])
    fe.detect(0, n)
    feats = [fe.features(s) for s in range(n)]
    pairs = [[0, 1], [1, 2], [2, 3], [3, 0]]
    res = _check_pairs(oracle, fe, feats, pairs, seq["K"], F.MATCH_CROSSCHECK, 2)
    assert res["n_inl"][:3].min() > 200
    from visual_odometry_amd import synth
    for p in range(3):
        Rgt, tgt = synth.relative_pose(seq["R"][p], seq["C"][p], seq["R"][p + 1], seq["C"][p + 1])
        assert np.linalg.norm(res["R"][p].reshape(3, 3) - Rgt) < 0.02 and abs(float(res["t"][p] @ tgt)) > 0.95


def test_sift_matcher_modes_and_small_frames(oracle, kernel_dk_rule):
    """Smaller frames (the whole pipeline at another geometry: other octave count, strips narrower than a workgroup's), every
    matcher mode, uneven keypoint counts."""
    from visual_odometry_amd import frontend as F
    w, h, n = 416, 240, 3
    seq = _flight(n, w, h)
    fe = F.FrontEnd(h, w, max_frames=n, max_pairs=n, detector="sift", kp_cap=2048)
    fe.upload(seq["frames"])
    fe.detect(0, n)
    feats = [fe.features(s) for s in range(n)]
    for s in range(n):
        _same_features(feats[s], oracle.sift_detect_and_compute(seq["frames"][s]))
    pairs = [[0, 1], [1, 2], [2, 0]]
    _check_pairs(oracle, fe, feats, pairs, seq["K"], F.MATCH_CROSSCHECK, 2)
    _check_pairs(oracle, fe, feats, pairs, seq["K"], F.MATCH_CROSSCHECK_LEGACY, 1)
    _check_pairs(oracle, fe, feats, pairs, seq["K"], F.MATCH_RATIO, "ratio", ratio=0.8)


def test_l2_matcher_on_both_sides_of_4096_rows(oracle, kernel_dk_rule):
    """k_nn_l2i8 keeps value and 16-row group in one 32-bit key while the train frame has at most 4096 rows (256 groups) and falls
    back to separate value / group registers above: a pair with 4415 keypoints per frame (contrastThreshold 0.025 at 1280x720)
    takes the second form in both directions — same pairs, distances, masks, E, R | t as the oracle."""
    from visual_odometry_amd import frontend as F
    w, h = 1280, 720
    seq = _flight(2, w, h)
    fe = F.FrontEnd(h, w, max_frames=2, max_pairs=1, detector="sift", kp_cap=8192, contrastThreshold=0.025)
    fe.upload(seq["frames"]); fe.detect(0, 2)
    feats = [fe.features(s) for s in range(2)]
    for s in range(2):
        _same_features(feats[s], oracle.sift_detect_and_compute(seq["frames"][s], contrast_threshold=0.025))
    assert min(len(f["xy"]) for f in feats) > 4096
    _check_pairs(oracle, fe, feats, [[0, 1]], seq["K"], F.MATCH_CROSSCHECK, 2)


def test_sift_batch_sub_batches_and_reuse(oracle):
    """More frames than a sub-batch holds (scratch reused), slots detected in two calls, then the same context reconfigured."""
    import os
    from visual_odometry_amd import frontend as F
    os.environ["VO_SIFT_SUBBATCH"] = "3"
    try:
        w, h, n = 320, 200, 7
        frames = np.stack([random_image(50 + i, h, w) for i in range(n)])
        fe = F.FrontEnd(h, w, max_frames=n, max_pairs=2, detector="sift", kp_cap=4096)
        fe.upload(frames)
        fe.detect(0, 5)
        fe.detect(5, 2)
        for s in (0, 2, 3, 4, 6):
            _same_features(fe.features(s), oracle.sift_detect_and_compute(frames[s]))
        fe2 = F.FrontEnd(97, 131, max_frames=2, max_pairs=1, detector="sift", ctx=fe.ctx, nOctaveLayers=4, sigma=1.4, contrastThreshold=0.03, edgeThreshold=8)
        img = random_image(7, 97, 131)
        fe2.upload(np.stack([img, img]))
        fe2.detect(0, 2)
        want = oracle.sift_detect_and_compute(img, n_layers=4, sigma=1.4, contrast_threshold=0.03, edge_threshold=8)
        _same_features(fe2.features(0), want); _same_features(fe2.features(1), want)
    finally:
        del os.environ["VO_SIFT_SUBBATCH"]


def test_sift_capacity_is_flagged(oracle):
    from visual_odometry_amd import frontend as F
    img = random_image(3, 240, 320)
    want = oracle.sift_detect_and_compute(img)
    assert want["n_found"] > 300
    fe = F.FrontEnd(240, 320, max_frames=2, max_pairs=1, detector="sift", kp_cap=256)
    fe.upload(np.stack([img, img]))
    with pytest.warns(RuntimeWarning, match="capacity"):                 # vo_frames_detect returns VO_WARN_CAPACITY ...
        fe.detect(0, 2)
    with pytest.warns(RuntimeWarning, match="capacity"):                 # ... and so does vo_pairs_run for a pair that involves the slot
        res, _ = fe.run_pairs([[0, 1]], np.array([[300.0, 0, 160], [0, 300.0, 120], [0, 0, 1]]))
    assert res["n_kp1"][0] == 256 and res["n_kp2"][0] == 256            # the kept count, not the uncapped one
    got = fe.features(0)
    assert got["truncated"] and len(got["xy"]) == 256
    for k in KEYS:                                                       # truncated in cv2's list order: a prefix of the full list
        assert np.array_equal(got[k], want[k][:256]), k
    assert np.array_equal(got["desc"], want["desc"][:256])


def test_image_and_keypoints_sift_branch(oracle):
    """ImageAndKeypoints takes SIFT + BFMatcher(NORM_L2, crossCheck=True) for any name but "ORB" (src/image_and_keypoints.py:10-13)."""
    from visual_odometry_amd import ImageAndKeypoints
    from visual_odometry_amd.detector import SiftDetector
    from visual_odometry_amd.matcher import L2Matcher
    iak = ImageAndKeypoints("SIFT")
    assert isinstance(iak.detector, SiftDetector) and isinstance(iak.bf, L2Matcher)
    g = random_image(21, 120, 160)
    iak.set_image(np.stack([g, g, g], axis=2))
    iak.detect_keypoints()
    want = oracle.sift_detect_and_compute(np.stack([g, g, g], axis=2))
    assert len(iak.keypoints) == want["n_found"] and np.array_equal(iak.descriptors, want["desc"])
    assert len(iak.kp_colors) == len(iak.keypoints)
    assert isinstance(ImageAndKeypoints("anything else").detector, SiftDetector)


def test_orb_and_sift_front_ends_share_a_context(oracle):
    """vo_batch_configure / vo_batch_configure_sift switch one context between the two detectors."""
    from visual_odometry_amd import frontend as F
    seq = _flight(2, 416, 240)
    fo = F.FrontEnd(240, 416, max_frames=2, max_pairs=1, nfeatures=300, nlevels=4)
    fo.upload(seq["frames"]); fo.detect(0, 2)
    orb0 = fo.features(0)
    fs = F.FrontEnd(240, 416, max_frames=2, max_pairs=1, detector="sift", ctx=fo.ctx, kp_cap=2048)
    fs.upload(seq["frames"]); fs.detect(0, 2)
    _same_features(fs.features(0), oracle.sift_detect_and_compute(seq["frames"][0]))
    fo2 = F.FrontEnd(240, 416, max_frames=2, max_pairs=1, nfeatures=300, nlevels=4, ctx=fo.ctx)
    fo2.upload(seq["frames"]); fo2.detect(0, 2)
    again = fo2.features(0)
    assert np.array_equal(again["xy"], orb0["xy"]) and np.array_equal(again["desc"], orb0["desc"])
    res, _ = fo2.run_pairs([[0, 1]], seq["K"])
    assert res["status"][0] == 0 and res["n_inl"][0] > 20


def test_sift_overlapped_contexts_are_deterministic():
    """Two SIFT contexts alternating chunks (asynchronous detect + pairs on their own streams) reproduce the synchronous
    single-context results bit for bit, chunk after chunk, features included: the candidate lists are filled through
    atomic counters in whatever order the waves arrive, the sort makes the keypoint order a function of the data alone, and the
    descriptor / orientation kernels rely on wave-level ordering of their LDS traffic instead of barriers."""
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd
    C = 20
    seq = synth.sequence(7, 640, 360, cache_dir="/tmp")
    order = [(i % 12 if i % 12 < 7 else 12 - i % 12) for i in range(C + 1)]
    frames = seq["frames"][order]
    K = seq["K"]
    pairs = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
    fes = [FrontEnd(360, 640, max_frames=C + 1, max_pairs=C, detector="sift") for _ in range(2)]
    for f in fes:
        f.upload(frames)

    def key(r):
        return np.concatenate([r[k].astype(np.float64).ravel() for k in ("status", "n_kp1", "n_match", "n_inl", "n_good", "ransac_iters", "R", "t", "E")])

    fes[0].detect(0, C + 1)
    ref = key(fes[0].run_pairs(pairs, K)[0])
    feat = [fes[0].features(sl) for sl in (0, 3, C)]
    assert min(len(f["xy"]) for f in feat) > 200
    inflight = [None, None]
    for it in range(10):
        k = it % 2
        if inflight[k] is not None:
            fes[k].wait()
            assert np.array_equal(key(inflight[k]), ref), f"chunk {it - 2} differs from the synchronous run"
        fes[k].detect(0, C + 1, wait=False)
        inflight[k] = fes[k].run_pairs(pairs, K, wait=False)[0]
    for k in range(2):
        fes[k].wait()
        assert np.array_equal(key(inflight[k]), ref)
        for sl, want in zip((0, 3, C), feat):
            got = fes[k].features(sl)
            for name in ("xy", "size", "angle", "response", "octave", "desc"):
                assert np.array_equal(got[name], want[name]), (k, sl, name)
