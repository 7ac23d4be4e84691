"""Oracle against the real cv2, stage by stage — runs only where `import cv2` works.

cv2 (opencv-python 4.7.0.72, Pipfile.lock:162-173) is neither in this image nor on the GPU box, so in the build
pipeline every test below SKIPS and parity stays "unpinned" (oracle/voo.h).  The module exists so that the first
environment that does have cv2 pins the oracle — and therefore the HIP path, which is bit-identical to the oracle —
with one command:  python -m pytest tests/test_cv2_crosscheck.py -q
The calls are the reference's own (frame_generator.py:25-26, image_pair.py:234-236, 280-286, 304-308, 332-336,
feature_detection.py:20-26), made directly from build-owned code on the build's synthetic inputs.
Integer stages must agree bit for bit; the float stages to the tolerances BASELINE.json states (R|t 1e-4, X 1e-3).
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from conftest import random_image


@pytest.fixture(scope="module")
def frames(seq_small):
    return seq_small["frames"], seq_small["K"]


def test_gray_matches_cvtcolor(oracle):
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    assert np.array_equal(oracle.gray(bgr), cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY))


@pytest.mark.parametrize("dst", [(533, 400), (107, 80), (639, 479)])
def test_resize_matches_inter_linear_exact(oracle, dst):
    img = random_image(11, 480, 640)
    ref = cv2.resize(img, dst, interpolation=cv2.INTER_LINEAR_EXACT)
    assert np.array_equal(oracle.resize_linear_exact(img, dst[0], dst[1]), ref)


@pytest.mark.parametrize("shape,dsize", [((216, 384, 3), (115, 64)), ((97, 131), (64, 48)), ((60, 80, 3), (40, 30))])
def test_ingest_resize_matches_cv2_resize(oracle, shape, dsize):
    # visual_slam.py:346-352; an IPP-enabled cv2 may differ by one grey level (oracle/voo_ingest.c), so report both
    rng = np.random.default_rng(31)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = cv2.resize(img, dsize)
    got = oracle.resize_linear(img, dsize[0], dsize[1])
    assert np.abs(ref.astype(int) - got.astype(int)).max() <= 1
    cv2.ipp.setUseIPP(False) if hasattr(cv2, "ipp") else None
    assert np.array_equal(cv2.resize(img, dsize), got)


def test_blur_matches_orbs_gaussianblur_path(oracle):
    """orb.cpp blurs every level as a SUB-MATRIX of the bordered pyramid buffer: GaussianBlur(workingMat, workingMat,
    Size(7, 7), 2, 2, BORDER_REFLECT_101) with workingMat = imagePyramid(layerInfo[level]).  For an 8-bit sub-matrix
    without BORDER_ISOLATED GaussianBlur skips its bit-exact fixed-point branch and runs sepFilter2D, whose 8-bit path
    quantises the float kernel to cvRound(256 g) = {18, 34, 49, 55, 49, 34, 18} and rounds (sum + 2^15) >> 16 — what
    the oracle restates.  A numpy view never carries cv::Mat's SUBMATRIX flag, so that branch cannot be reached through
    GaussianBlur from Python: the same arithmetic is called directly (sepFilter2D with GaussianBlur's own CV_32F
    kernel).  The pixels sepFilter2D reads beyond the ROI are the pyramid's REFLECT_101 border, i.e. the reflection."""
    img = random_image(12, 240, 320)
    g = cv2.getGaussianKernel(7, 2, cv2.CV_32F)
    ref = cv2.sepFilter2D(img, -1, g, g, borderType=cv2.BORDER_REFLECT_101)
    assert np.array_equal(oracle.gaussian_blur7(img), ref)
    # the same blur on a whole (non-sub-matrix) image takes the fixed-point branch with a different tap table: it is
    # expected to DIFFER from the oracle in a few grey levels, and must never be used to "pin" the ORB blur
    whole = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    diff = np.abs(whole.astype(int) - ref.astype(int))
    print(f"whole-image GaussianBlur vs sepFilter2D path: {int((diff > 0).sum())} pixels differ, max {int(diff.max())}")
    assert diff.max() <= 1


def test_fast_matches_fastfeaturedetector(oracle):
    img = random_image(13, 240, 320)
    det = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
    kps = det.detect(img, None)
    ref = np.zeros(img.shape, np.uint8)
    for kp in kps:
        ref[int(kp.pt[1]), int(kp.pt[0])] = int(kp.response)
    assert np.array_equal(oracle.fast_score_nms(img, 20), ref)


@pytest.mark.parametrize("nfeatures", [500, 2000])
def test_orb_keypoints_and_descriptors(oracle, frames, nfeatures):
    img = frames[0][0]
    orb = cv2.ORB_create(nfeatures=nfeatures)
    kps, desc = orb.detectAndCompute(img, None)
    got = oracle.orb_detect_and_compute(img, oracle.orb_params(nfeatures=nfeatures))
    ref = {(round(k.pt[0], 3), round(k.pt[1], 3), k.octave): (k.angle, k.response, d) for k, d in zip(kps, desc)}
    mine = {(round(float(x), 3), round(float(y), 3), int(o)): (a, r, d)
            for (x, y), o, a, r, d in zip(got["xy"], got["octave"], got["angle"], got["response"], got["desc"])}
    assert set(ref) == set(mine)                           # same keypoint SET (cv2's order is not canonical)
    for key, (a, r, d) in ref.items():
        a2, r2, d2 = mine[key]
        assert abs(a - a2) < 1e-3 and abs(r - r2) <= 1e-6 * max(1.0, abs(r))
        assert np.array_equal(d, d2)


def _descs(oracle, frames):
    f, _ = frames
    p = oracle.orb_params(nfeatures=500)
    return oracle.orb_detect_and_compute(f[0], p), oracle.orb_detect_and_compute(f[1], p)


def test_bfmatcher_crosscheck(oracle, frames):
    a, b = _descs(oracle, frames)
    ms = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(a["desc"], b["desc"])
    qi, ti, d = oracle.match_hamming(a["desc"], b["desc"], 2)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in ms] == list(zip(qi.tolist(), ti.tolist(), d.tolist()))


def test_crosscheck_rule_probe(oracle):
    """Which cross-check rule does this cv2 implement?  The oracle's default (2) is OpenCV 4.x's batchDistance: train i
    is kept for its nearest query idx only if idx's own nearest train row is i (mutual nearest neighbours).  Rule 1 is
    the older update without that forward test.  The two differ on a NON-mutual configuration without any tie."""
    q = np.zeros((2, 32), np.uint8); t = np.zeros((2, 32), np.uint8)
    q[1, 0] = 0b1111; t[0, 0] = 0b1; t[1, 0] = 0xff; t[1, 1] = 0b1
    # distances: q0-t0 1, q0-t1 9, q1-t0 3, q1-t1 5.  Reverse NN: t0 -> q0, t1 -> q1.  Forward NN: q0 -> t0, q1 -> t0.
    # rule 1 keeps (q0, t0) and (q1, t1); the mutual rule keeps only (q0, t0).
    got = [(m.queryIdx, m.trainIdx) for m in cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(q, t)]
    r1 = list(zip(*[a.tolist() for a in oracle.match_hamming(q, t, 1)[:2]]))
    r2 = list(zip(*[a.tolist() for a in oracle.match_hamming(q, t, 2)[:2]]))
    assert r1 == [(0, 0), (1, 1)] and r2 == [(0, 0)]
    assert got in (r1, r2), got
    assert got == r2, "this cv2 build uses the LEGACY cross-check rule (oracle mode 1): switch the default (vo_pair_opts.match_mode 2)"
    # ties: equal distances resolve to the lowest index in both directions
    rng = np.random.default_rng(3)
    tt = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    qq = np.repeat(tt[:10], 2, axis=0)                      # every query twice, every matching train row once
    tt[20:30] = tt[:10]                                     # and every matching train row twice
    ms = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(qq, tt)
    qi, ti, d = oracle.match_hamming(qq, tt, 2)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in ms] == list(zip(qi.tolist(), ti.tolist(), d.tolist()))


def test_knn_ratio(oracle, frames):
    a, b = _descs(oracle, frames)
    knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(a["desc"], b["desc"], k=2)
    ref = [(m.queryIdx, m.trainIdx, m.distance) for m, n in knn if m.distance < 0.8 * n.distance]
    qi, ti, d = oracle.knn2_ratio_hamming(a["desc"], b["desc"], 0.8)
    assert ref == list(zip(qi.tolist(), ti.tolist(), d.tolist()))


def _matched_points(oracle, frames):
    a, b = _descs(oracle, frames)
    qi, ti, _ = oracle.match_hamming(a["desc"], b["desc"], 2)
    return a["xy"][qi].astype(np.float64), b["xy"][ti].astype(np.float64)


def test_find_essential_mat(oracle, frames):
    p1, p2 = _matched_points(oracle, frames)
    K = frames[1]
    E, mask = cv2.findEssentialMat(p1, p2, K, cv2.FM_RANSAC, 0.99, 1)
    rc, Es, m, n = oracle.find_essential_ransac(p1, p2, K)
    assert rc == 0
    assert np.array_equal(mask.ravel(), m)                 # same RNG stream, same samples, same inlier set
    E0 = Es[0]
    assert min(np.abs(E - E0).max(), np.abs(E + E0).max()) < 1e-6


def test_recover_pose_and_triangulate(oracle, frames):
    p1, p2 = _matched_points(oracle, frames)
    K = frames[1]
    E, mask = cv2.findEssentialMat(p1, p2, K, cv2.FM_RANSAC, 0.99, 1)
    inl = mask.ravel() == 1
    q1, q2 = p1[inl], p2[inl]
    n_ref, R, t, pm = cv2.recoverPose(E, q1, q2, K)
    n, R2, t2, pm2 = oracle.recover_pose(E, q1, q2, K)
    assert n == n_ref and np.array_equal((pm.ravel() > 0), (pm2 > 0))
    assert np.abs(R - R2).max() < 1e-4 and np.abs(t - t2).max() < 1e-4
    P0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P1 = K @ np.hstack([R.T, -R.T @ t])                    # image_pair.py:319-328
    X = cv2.triangulatePoints(P1, P0, q1.T, q2.T)
    X2 = oracle.triangulate(P1, P0, q1.T, q2.T)
    X = X / X[3]; X2 = X2 / X2[3]
    near = np.linalg.norm(X[:3], axis=0) < 10 * np.median(np.linalg.norm(X[:3], axis=0))
    assert np.abs(X[:3, near] - X2[:3, near]).max() < 1e-3 * np.abs(X[:3, near]).max()


def test_solve_pnp_ransac_and_rodrigues(oracle):
    """visual_slam.py:231-243.  Tolerance only: cv2's EPnP takes its 12x12 SVD from LAPACK (oracle/voo_pnp.c)."""
    rng = np.random.default_rng(77)
    K = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
    r_true = np.array([0.1, -0.3, 0.2]); t_true = np.array([0.2, -0.1, 5.5])
    X = rng.uniform(-2, 2, (240, 3)); Xc = X @ cv2.Rodrigues(r_true)[0].T + t_true
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.4, (240, 2))
    bad = rng.random(240) < 0.35
    uv[bad] += rng.uniform(-80, 80, (int(bad.sum()), 2))
    ok, rvec, tvec, inl = cv2.solvePnPRansac(X, uv, K, np.zeros(4))
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
    assert ok and rc == 0
    ref = np.zeros(240, bool); ref[inl.ravel()] = True
    assert (ref == (mask > 0)).mean() > 0.98                    # threshold-edge points may flip with the hypothesis noise
    assert np.abs(rvec.ravel() - rv).max() < 1e-3 and np.abs(tvec.ravel() - tv).max() < 1e-2
    if np.array_equal(ref, mask > 0):                            # the same consensus set: the final solvePnP(ITERATIVE) is restated step by
        assert np.abs(rvec.ravel() - rv).max() < 1e-6 and np.abs(tvec.ravel() - tv).max() < 1e-6     # step (DLT start, CvLevMarq): only LAPACK's SVDs differ
    assert np.abs(cv2.Rodrigues(rv)[0] - oracle.rodrigues(rv)).max() < 1e-12
    # a planar map takes cvFindExtrinsicCameraParams2's homography start
    Xp = X.copy(); Xp[:, 2] = 0.2 * Xp[:, 0] - 0.1 * Xp[:, 1] + 0.5
    Xc = Xp @ cv2.Rodrigues(r_true)[0].T + t_true
    uvp = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.3, (240, 2))
    ok, rvec, tvec, inl = cv2.solvePnPRansac(Xp, uvp, K, np.zeros(4))
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(Xp, uvp, K)
    assert ok and rc == 0
    assert np.abs(rvec.ravel() - rv).max() < 1e-3 and np.abs(tvec.ravel() - tv).max() < 1e-2


def test_imdecode_and_sift(oracle):
    """cv2.imdecode on JPEG files (libjpeg-turbo as cv2 ships it: the far-out-of-range golden file included) and
    cv2.SIFT_create().detectAndCompute + BFMatcher(NORM_L2, crossCheck=True): the rows SURVEY 8(f) ranks next."""
    import os
    buf = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_gray_q1_saturated_329x267.jpg"), "rb").read()
    assert np.array_equal(oracle.jpeg_decode(buf), cv2.imdecode(np.frombuffer(buf, np.uint8), cv2.IMREAD_COLOR))
    img = random_image(5, 240, 320)
    for q, ss in ((90, cv2.IMWRITE_JPEG_SAMPLING_FACTOR_420 if hasattr(cv2, "IMWRITE_JPEG_SAMPLING_FACTOR_420") else None), (35, None)):
        params = [cv2.IMWRITE_JPEG_QUALITY, q] + ([cv2.IMWRITE_JPEG_SAMPLING_FACTOR, ss] if ss is not None else [])
        ok, enc = cv2.imencode(".jpg", np.stack([img, np.roll(img, 3, 1), np.roll(img, 5, 0)], -1), params)
        assert ok and np.array_equal(oracle.jpeg_decode(enc.tobytes()), cv2.imdecode(enc, cv2.IMREAD_COLOR))
    kps, desc = cv2.SIFT_create().detectAndCompute(img, None)
    want = oracle.sift_detect_and_compute(img)
    assert len(kps) == want["n_found"]
    assert np.array_equal(np.array([k.pt for k in kps], np.float32), want["xy"]) and np.array_equal(desc, want["desc"])
    assert np.array_equal(np.array([k.octave for k in kps]), want["octave"]) and np.array_equal(np.array([k.angle for k in kps], np.float32), want["angle"])
    img2 = np.roll(img, 4, 1)
    d2 = cv2.SIFT_create().detectAndCompute(img2, None)[1]
    ms = cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match(desc, d2)
    q, t, d = oracle.match_l2(desc, d2, 2)
    assert [m.queryIdx for m in ms] == q.tolist() and [m.trainIdx for m in ms] == t.tolist() and np.array_equal(np.array([m.distance for m in ms], np.float32), d)
