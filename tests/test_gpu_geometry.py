"""GPU parity: five-point solver, essential-matrix RANSAC, recoverPose and DLT triangulation against the CPU
oracle.  The north star asks [R|t] within 1e-4 (Frobenius) and points within 1e-3 relative; kernel and oracle run
the same IEEE operations in the same order, so models, masks, E, R and t are asserted bit for bit here (the oracle's
root finder set to the kernel's exit rule, see conftest.kernel_dk_rule); tests/test_gpu_faithful.py holds the
comparisons against cv::solvePoly's fixed 300 sweeps."""
import numpy as np
import pytest
from twoview import fuzz_problem, five_point_sample, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _same_root_finder_rule(kernel_dk_rule):
    """Every test of this module compares the product's DEFAULT mode with the oracle bit for bit, so the oracle runs the
    kernel's Durand-Kerner exit rule; tests/test_gpu_faithful.py covers OpenCV's fixed 300 sweeps on both sides."""
    yield


def test_five_point_models_match_oracle(oracle, ctx):
    from visual_odometry_amd import geometry
    rng = np.random.default_rng(0)
    for _ in range(40):
        x1, x2 = five_point_sample(rng)
        got, ref = geometry.five_point(x1, x2), oracle.five_point(x1, x2)
        assert got.shape == ref.shape                    # same number of real roots, same order
        assert np.array_equal(got, ref)                  # same IEEE operations in the same order: bit identical
        for E in got:                                     # the defining constraints hold on the GPU result
            assert max(abs(np.r_[x2[i], 1] @ E @ np.r_[x1[i], 1]) for i in range(5)) < 1e-8
            assert abs(np.linalg.det(E)) < 1e-6       # a sample the root finder gave up on (64 sweeps) sits at ~1e-8


@pytest.mark.parametrize("seed,n,outl", [(1, 800, 0.3), (2, 2000, 0.2), (3, 300, 0.5), (4, 64, 0.1), (5, 1500, 0.7)])
def test_find_essential_matches_oracle(oracle, ctx, seed, n, outl):
    from visual_odometry_amd import geometry
    K, R, t, p1, p2 = scene(seed, n, outliers=outl)
    rc, Er, mr, nr = oracle.find_essential_ransac(p1, p2, K)
    E, mask = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1)
    assert rc == 0 and E is not None and E.shape == (3, 3) and mask.shape == (n, 1)
    assert np.array_equal(mask.ravel(), mr)              # identical inlier set
    assert np.array_equal(E, Er[0])                      # same winning model, bit for bit


def test_recover_pose_and_triangulate_match_oracle(oracle, ctx):
    from visual_odometry_amd import geometry
    for seed in (11, 12, 13):
        K, R, t, p1, p2 = scene(seed, 900)
        E, mask = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1)
        q1, q2 = p1[mask.ravel() == 1], p2[mask.ravel() == 1]
        ng, Rg, tg, mg = geometry.recoverPose(E, q1, q2, K)
        nr, Rr, tr, mr = oracle.recover_pose(E, q1, q2, K)
        assert ng == nr and np.array_equal(mg.ravel(), mr)
        assert np.array_equal(Rg, Rr) and np.array_equal(tg, tr)
        assert abs(np.linalg.norm(tg) - 1) < 1e-12 and abs(np.linalg.det(Rg) - 1) < 1e-9
        P1 = K @ np.hstack([Rg.T, -Rg.T @ tg]); P0 = K @ np.eye(3, 4)
        Xg = geometry.triangulatePoints(P1, P0, q1.T, q2.T)
        Xr = oracle.triangulate(P1, P0, q1.T, q2.T)
        Xg /= Xg[3]; Xr /= Xr[3]
        rel = np.linalg.norm(Xg[:3] - Xr[:3], axis=0) / np.linalg.norm(Xr[:3], axis=0)
        assert rel.max() < 1e-3


def test_pose_close_to_ground_truth(ctx):
    from visual_odometry_amd import geometry
    K, R, t, p1, p2 = scene(21, 1200, noise=0.1, outliers=0.2)
    E, mask = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1)
    _, Rg, tg, _ = geometry.recoverPose(E, p1[mask.ravel() == 1], p2[mask.ravel() == 1], K)
    assert np.linalg.norm(Rg - R) < 0.05 and np.linalg.norm(tg.ravel() - t) < 0.1


def test_degenerate_inputs(oracle, ctx):
    from visual_odometry_amd import geometry
    K, R, t, p1, p2 = scene(31, 200)
    assert geometry.findEssentialMat(p1[:4], p2[:4], K) == (None, None)      # cv2 returns None below 5 points
    E, mask = geometry.findEssentialMat(p1[:5], p2[:5], K, geometry.FM_RANSAC, 0.99, 1)   # exactly 5: stacked models
    rc, Er, mr, nr = oracle.find_essential_ransac(p1[:5], p2[:5], K)
    assert E.shape[0] % 3 == 0 and E.shape == (3 * len(Er), 3) and np.all(mask == 1)
    assert np.array_equal(E.reshape(-1, 3, 3), Er)
    X = geometry.triangulatePoints(K @ np.eye(3, 4), K @ np.eye(3, 4), np.zeros((2, 0)), np.zeros((2, 0)))
    assert X.shape == (4, 0)


def test_fixed_seed_is_deterministic(ctx):
    from visual_odometry_amd import geometry
    K, R, t, p1, p2 = scene(41, 700)
    a = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1)
    b = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c = geometry.findEssentialMat(p1, p2, K, geometry.FM_RANSAC, 0.99, 1, seed=12345)
    assert c[0] is not None


@pytest.mark.parametrize("seed", range(6))
def test_randomised_two_view_problems(oracle, ctx, seed):
    """Differential fuzz of findEssentialMat -> recoverPose -> triangulatePoints: random motions (incl. pure rotation and
    pure forward motion), planar and deep scenes, 8..1500 points, 0..85 % outliers, random intrinsics and thresholds."""
    rng = np.random.default_rng(500 + seed)
    for _ in range(5):
        pr = fuzz_problem(rng)
        K, p1, p2, thresh, prob = pr["K"], pr["p1"], pr["p2"], pr["thresh"], pr["prob"]
        rc, Es, mask, ninl = oracle.find_essential_ransac(p1, p2, K, prob=prob, thresh=thresh)
        E, m = geometry().findEssentialMat(p1, p2, K, prob=prob, threshold=thresh)
        tag = f"seed {seed} " + pr["tag"]
        if rc != 0:
            assert E is None, tag
            continue
        assert np.array_equal(m.ravel(), mask) and np.array_equal(E, Es[0]), tag
        inl = mask > 0
        ng, Rr, tr, pm = oracle.recover_pose(Es[0], p1[inl], p2[inl], K)
        ng2, R2, t2, pm2 = geometry().recoverPose(E, p1[inl], p2[inl], K)
        assert ng2 == ng and np.array_equal(pm2.ravel() > 0, pm > 0), tag
        assert np.array_equal(R2, Rr) and np.array_equal(t2, tr), tag


def geometry():
    from visual_odometry_amd import geometry as g
    return g
