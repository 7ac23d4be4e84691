"""SIFT on the MI355X (the reference's live detector, /root/reference/src/visual_slam.py:17) against the CPU oracle
(oracle/voo_sift.c): keypoints (position, size, angle, response, packed octave) and the 128-float descriptors, bit for bit;
then the reference's live pair path: SIFT + BFMatcher(NORM_L2, crossCheck=True) through ImagePair."""
import numpy as np
import pytest

from conftest import random_image

pytestmark = pytest.mark.gpu


def _check(oracle, ctx, img, **kw):
    from visual_odometry_amd.detector import SiftDetector
    det = SiftDetector(ctx=ctx, **kw)
    got = det.detect_arrays(img)
    want = oracle.sift_detect_and_compute(img, n_layers=kw.get("nOctaveLayers", 3), contrast_threshold=kw.get("contrastThreshold", 0.04),
                                          edge_threshold=kw.get("edgeThreshold", 10.0), sigma=kw.get("sigma", 1.6), nfeatures=kw.get("nfeatures", 0))
    assert len(got["xy"]) == want["n_found"] and want["n_found"] > 0
    for key in ("xy", "size", "angle", "response", "octave"):
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(got["desc"], want["desc"])
    return got


@pytest.mark.parametrize("h,w", [(64, 64), (97, 131), (240, 320), (123, 457), (480, 640)])
def test_sift_equals_oracle(oracle, ctx, h, w):
    got = _check(oracle, ctx, random_image(h * 7 + w, h, w))
    n = np.linalg.norm(got["desc"], axis=1)
    assert got["desc"].min() >= 0 and got["desc"].max() <= 255 and np.all(np.abs(n - 512) < 40)


def test_sift_golden_vectors(ctx):
    """The committed fixture (tests/golden/sift_96x128.npz, written by the oracle): needs nothing but the HIP library."""
    import os
    from visual_odometry_amd.detector import SiftDetector
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sift_96x128.npz"))
    r = SiftDetector(ctx=ctx).detect_arrays(g["img"])
    assert len(r["xy"]) == int(g["n"])
    for k in ("xy", "size", "angle", "response", "octave"):
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(r["desc"].astype(np.uint8), g["desc"])


def test_sift_colour_input_and_parameters(oracle, ctx):
    g = random_image(11, 200, 260)
    bgr = np.stack([g, np.roll(g, 2, 0), np.roll(g, 3, 1)], axis=2)
    _check(oracle, ctx, bgr)
    _check(oracle, ctx, g, nOctaveLayers=4, contrastThreshold=0.03, edgeThreshold=8, sigma=1.4)
    _check(oracle, ctx, g, nOctaveLayers=2)
    for nf in (1, 50, 400, 100000):                                       # retainBest: cv2's permutation of the sorted list, ties kept
        got = _check(oracle, ctx, g, nfeatures=nf)
        assert len(got["xy"]) >= min(nf, 1)


def test_sift_smooth_and_flat_images(oracle, ctx):
    from visual_odometry_amd.detector import SiftDetector
    yy, xx = np.mgrid[0:160, 0:200]
    smooth = (127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    _check(oracle, ctx, smooth)
    flat = np.full((80, 90), 77, np.uint8)
    got = SiftDetector(ctx=ctx).detect_arrays(flat)
    assert len(got["xy"]) == 0 and oracle.sift_detect_and_compute(flat)["n_found"] == 0


def test_live_pair_path_sift_l2(oracle, seq_small):
    """visual_slam.py:17-21,294-298 as the reference runs it: SIFT features, L2 cross-check matcher, E-RANSAC, pose."""
    import visual_odometry_amd as vo
    from visual_odometry_amd.frame_generator import FrameGenerator
    from visual_odometry_amd.image_pair import ImagePair
    frames, K = seq_small["frames"], seq_small["K"]
    gen = FrameGenerator(vo.SIFT_create())
    bf = vo.BFMatcher(vo.NORM_L2, crossCheck=True)
    f1, f2 = gen.make_frame(frames[0]), gen.make_frame(frames[1])
    o1, o2 = oracle.sift_detect_and_compute(frames[0]), oracle.sift_detect_and_compute(frames[1])
    assert np.array_equal(f1.descriptors, o1["desc"]) and np.array_equal(f2.descriptors, o2["desc"])
    ip = ImagePair(f1, f2, bf, K)
    ip.match_features()
    qi, ti, dd = oracle.match_l2(o1["desc"], o2["desc"], 2)
    assert [m.featureid1[1] for m in ip.raw_matches] == qi.tolist() and [m.featureid2[1] for m in ip.raw_matches] == ti.tolist()
    ess = ip.determine_essential_matrix(ip.filtered_matches)
    assert len(ess) > 50
    p1, p2 = ip.get_image_points(ip.filtered_matches)
    rc, E, mask, ninl = oracle.find_essential_ransac(p1, p2, K)              # the same RANSAC on the CPU: same inlier set
    assert rc == 0 and ninl == len(ess)
    ip.estimate_camera_movement(ess)
    from visual_odometry_amd import synth
    Rgt, tgt = synth.relative_pose(seq_small["R"][0], seq_small["C"][0], seq_small["R"][1], seq_small["C"][1])
    assert np.linalg.norm(ip.R - Rgt) < 0.05 and abs(float(ip.t.ravel() @ tgt)) > 0.9   # (cross-check only, no ratio test: a loose sanity bound)


def test_sift_differential_fuzz(oracle, ctx):
    """Random sizes (odd, thin, tiny octave pyramids), textures and parameters — tap counts other than cv2's default set go through the
    run-time form of the sweep kernel, odd widths through its single-column stores and the base kernel's tail."""
    from visual_odometry_amd.detector import SiftDetector
    rng = np.random.default_rng(2026)
    found = 0
    for case in range(14):
        h, w = int(rng.integers(24, 220)), int(rng.integers(24, 260))
        kind = case % 3
        if kind == 0:
            img = random_image(1000 + case, h, w)
        elif kind == 1:                                                    # blobs on a gradient
            yy, xx = np.mgrid[0:h, 0:w]
            img = 40.0 + 0.4 * xx + 0.2 * yy
            for _ in range(25):
                cy, cx, sg = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(1.5, 7)
                img = img + rng.uniform(-90, 90) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
            img = np.clip(img, 0, 255).astype(np.uint8)
        else:                                                              # coarse noise, up-sampled: corners at many scales
            small = rng.integers(0, 256, (h // 5 + 2, w // 5 + 2)).astype(np.uint8)
            img = np.kron(small, np.ones((5, 5), np.uint8))[:h, :w].copy()
        kw = dict(nOctaveLayers=int(rng.integers(2, 6)), contrastThreshold=float(rng.uniform(0.02, 0.07)),
                  edgeThreshold=float(rng.uniform(5, 15)), sigma=float(rng.uniform(1.2, 2.1)))
        want = oracle.sift_detect_and_compute(img, n_layers=kw["nOctaveLayers"], contrast_threshold=kw["contrastThreshold"],
                                              edge_threshold=kw["edgeThreshold"], sigma=kw["sigma"])
        got = SiftDetector(ctx=ctx, **kw).detect_arrays(img)
        assert len(got["xy"]) == want["n_found"], (case, h, w, kw)
        for key in ("xy", "size", "angle", "response", "octave"):
            assert np.array_equal(got[key], want[key]), (case, key)
        assert np.array_equal(got["desc"], want["desc"]), case
        found += want["n_found"]
    assert found > 500


def test_contrast_threshold_that_rounds_to_zero(oracle, ctx):
    """tests/golden/sift_flat_threshold0_265x354.npz (found by tests/scripts/soak_sift.py): contrastThreshold 0.0156 with five layers
    per octave makes the extrema threshold floor(0.5 * 0.0156 / 5 * 255) = 0, so every sample of the flat regions is a scale-space
    "extremum" (>= comparisons) — 1.9 million candidates of which the refinement keeps 57.  The per-image call's candidate list
    holds every DoG sample (cv2's lists are unbounded); with nfeatures = 50 retainBest keeps 51 (a tie)."""
    import os
    from visual_odometry_amd.detector import SiftDetector
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sift_flat_threshold0_265x354.npz"))
    kw = dict(nOctaveLayers=int(d["nOctaveLayers"]), contrastThreshold=float(d["contrastThreshold"]), edgeThreshold=float(d["edgeThreshold"]),
              sigma=float(d["sigma"]))
    for nf, count in ((0, 57), (50, 51)):
        want = oracle.sift_detect_and_compute(d["img"], nfeatures=nf, n_layers=kw["nOctaveLayers"], contrast_threshold=kw["contrastThreshold"],
                                              edge_threshold=kw["edgeThreshold"], sigma=kw["sigma"])
        got = SiftDetector(nfeatures=nf, ctx=ctx, **kw).detect_arrays(d["img"])
        assert want["n_found"] == count and len(got["xy"]) == count and not got["truncated"]
        for key in ("xy", "size", "angle", "response", "octave", "desc"):
            assert np.array_equal(got[key], want[key]), (nf, key)
