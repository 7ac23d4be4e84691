"""GPU parity of the localisation row: cv2.solvePnPRansac + cv2.Rodrigues
(src/visual_slam.py:231-243) against the CPU oracle — identical inlier sets, poses to POSE_TOL: cv2's final Levenberg-Marquardt
stops when a step falls below FLT_EPSILON (relative), so a rounding-level difference (libm's acos / cos / sin; the device takes the
dot products of its Jacobi SVDs as butterfly sums over lanes) can move the stop by one iteration, i.e. by ~1e-9; everything
before that — hypotheses, float32 errors, inlier sets — is compared exactly.  Default mode on both sides: the final pose as cv2 computes it (solvePnP(ITERATIVE) on
the consensus set: DLT or homography start + CvLevMarq); the fast mode is held against the oracle's and against it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = np.array([[800., 0, 320], [0, 800, 240], [0, 0, 1]])
POSE_TOL = 1e-7


def problem(seed, n, outl, noise=0.5, depth=6.0):
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = rng.uniform(0.1, 0.6)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
    t = np.array([0.3, -0.2, depth]) + rng.normal(0, 0.3, 3)
    X = rng.uniform(-2, 2, (n, 3))
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outl
    uv[bad] += rng.uniform(-100, 100, (int(bad.sum()), 2))
    return X, uv, R, t, bad


@pytest.mark.parametrize("seed,n,outl", [(1, 200, 0.3), (2, 50, 0.1), (3, 1000, 0.5), (4, 12, 0.0), (5, 6, 0.0), (6, 5, 0.0),
                                         (7, 2000, 0.6), (8, 300, 0.8)])
def test_solve_pnp_ransac_matches_oracle(oracle, ctx, seed, n, outl):
    from visual_odometry_amd import geometry
    X, uv, R, t, bad = problem(seed, n, outl)
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
    ok, rvec, tvec, inl = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
    assert ok == (rc == 0)
    if not ok:
        return
    assert np.array_equal(inl.ravel(), np.nonzero(mask)[0])            # same hypotheses, same float32 errors
    assert np.abs(rvec.ravel() - rv).max() < POSE_TOL and np.abs(tvec.ravel() - tv).max() < POSE_TOL
    Rm, _ = geometry.Rodrigues(rvec)
    if outl <= 0.6:                                                     # and it is the generating pose, to the noise level
        assert np.abs(Rm - R).max() < 0.02 and np.abs(tvec.ravel() - t).max() < 0.1
    if outl <= 0.5:                                                     # (100 iterations are few for more outliers: the
        assert ((mask > 0) == ~bad).mean() > 0.97                       #  mask is the best HYPOTHESIS' mask, as in cv2)


def test_batch_equals_single_calls(oracle, ctx):
    from visual_odometry_amd import geometry
    probs = [problem(20 + k, n, o) for k, (n, o) in enumerate([(150, 0.2), (5, 0.0), (700, 0.4), (3, 0.0), (4, 0.0), (64, 0.3)])]
    obj = np.concatenate([p[0] for p in probs]); img = np.concatenate([p[1] for p in probs])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).astype(np.int32)
    status, rvec, tvec, mask, ninl = geometry.solve_pnp_ransac_batch(obj, img, off, K)
    for b, p in enumerate(probs):
        rc, rv, tv, m, n_in = oracle.solve_pnp_ransac(p[0], p[1], K)
        assert status[b] == rc
        if rc == 0:
            assert ninl[b] == n_in and np.array_equal(mask[off[b]:off[b + 1]], m)
            assert np.abs(rvec[b] - rv).max() < POSE_TOL and np.abs(tvec[b] - tv).max() < POSE_TOL


def test_rodrigues_and_error_paths(oracle, ctx):
    from visual_odometry_amd import _lib, geometry
    rng = np.random.default_rng(5)
    for _ in range(20):
        r = rng.normal(0, 1.2, 3)
        Rm, _ = geometry.Rodrigues(r)
        assert np.abs(Rm - oracle.rodrigues(r)).max() < 1e-14 and np.abs(Rm @ Rm.T - np.eye(3)).max() < 1e-14
        back, _ = geometry.Rodrigues(Rm)
        assert np.abs(back.ravel() - oracle.rodrigues(Rm)).max() < 1e-12
    assert np.array_equal(geometry.Rodrigues(np.zeros(3))[0], np.eye(3))
    X, uv, *_ = problem(9, 3, 0.0)
    with pytest.raises(_lib.VoError):
        geometry.solvePnPRansac(X, uv, K, np.zeros(4))                  # cv2 asserts npoints >= 4
    X, uv, *_ = problem(10, 40, 0.0)
    ok, *_ = geometry.solvePnPRansac(X, np.random.default_rng(1).uniform(0, 600, uv.shape), K, np.zeros(4))
    assert ok in (True, False)                                          # garbage correspondences: a verdict, no crash
    with pytest.raises(NotImplementedError):
        geometry.solvePnPRansac(X, uv, K, np.array([0.1, 0, 0, 0]))


def test_degenerate_configurations_terminate_and_agree(oracle, ctx):
    """Coincident, collinear and coplanar object points, points behind the camera: no hang, and the same verdict and
    inlier set as the oracle (NaN hypotheses simply collect no inliers)."""
    from visual_odometry_amd import geometry
    rng = np.random.default_rng(4)
    uv = rng.uniform(0, 600, (60, 2))
    same = np.tile(np.array([[0.5, -0.2, 4.0]]), (60, 1))
    line = np.outer(np.linspace(-1, 1, 60), [1.0, 0.5, 0.2]) + [0, 0, 5]
    X, uvp, R, t, _ = problem(31, 60, 0.2)
    plane = X.copy(); plane[:, 2] = 0.0
    Xc = plane @ R.T + t
    uv_plane = ((Xc / Xc[:, 2:]) @ K.T)[:, :2]
    behind = X.copy(); behind[:, 2] -= 20.0
    for obj, img in ((same, uv), (line, uv), (plane, uv_plane), (behind, uvp)):
        rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(obj, img, K)
        st, rvec, tvec, m, n_in = geometry.solve_pnp_ransac_batch(obj, img, np.array([0, len(obj)], np.int32), K)
        assert st[0] == rc
        if rc == 0:
            assert n_in[0] == ninl and np.array_equal(m, mask)
            ok = np.isfinite(rv).all() and np.isfinite(tv).all()
            assert np.isfinite(rvec[0]).all() == ok
            if ok:
                assert np.abs(rvec[0] - rv).max() < 1e-6 and np.abs(tvec[0] - tv).max() < 1e-6


def test_four_points_take_the_p3p_branch(oracle, ctx):
    """cv2.solvePnPRansac with exactly four correspondences (visual_slam.py:231-235 with four matched map points):
    model_points == npoints, one solvePnP(SOLVEPNP_P3P) call, all four points inliers.  Same operations as the oracle
    (Gao's P3P, Ferrari quartic, Horn alignment); acos / cos / pow come from different math libraries: 1e-9."""
    from visual_odometry_amd import geometry
    rng = np.random.default_rng(21)
    exact = 0
    for it in range(60):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 1.0)
        kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * kx @ kx
        t = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4, 8)])
        X = rng.uniform(-2, 2, (4, 3))
        Xc = X @ R.T + t
        uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + (rng.normal(0, 0.3, (4, 2)) if it % 2 else 0)
        rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
        ok, rvec, tvec, inl = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
        assert ok == (rc == 0)
        if not ok:
            continue
        assert inl.ravel().tolist() == [0, 1, 2, 3] and ninl == 4
        assert np.abs(rvec.ravel() - rv).max() < 1e-9 and np.abs(tvec.ravel() - tv).max() < 1e-9
        if it % 2 == 0:                                      # noise-free: the true pose is among P3P's candidates and wins
            exact += np.abs(geometry.Rodrigues(rvec)[0] - R).max() < 1e-4 and np.abs(tvec.ravel() - t).max() < 1e-3
    assert exact >= 27                                        # float32 image points; a rare ambiguous configuration may pick a twin


def planar_problem(seed, n, outl, noise=0.5):
    X, uv, R, t, bad = problem(seed, n, 0.0, noise=0.0)
    rng = np.random.default_rng(1000 + seed)
    X[:, 2] = 0.3 * X[:, 0] - 0.2 * X[:, 1] + 1.0                      # all map points in one plane: cv2 starts from a homography
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outl
    uv[bad] += rng.uniform(-100, 100, (int(bad.sum()), 2))
    return X, uv, R, t, bad


@pytest.fixture()
def refine_modes(oracle, ctx):
    def set_mode(m):
        oracle.set_pnp_refine(m); ctx.set_pnp_refine(m)
    yield set_mode
    set_mode("cv2")


@pytest.mark.parametrize("seed,n,outl", [(41, 300, 0.3), (42, 40, 0.1), (43, 9, 0.0), (44, 6, 0.0), (45, 1200, 0.5)])
def test_planar_structure_takes_the_homography_start(oracle, ctx, seed, n, outl):
    """cvFindExtrinsicCameraParams2's planar branch (W[2] / W[1] < 1e-3): findHomography (normalised DLT + LMSolver) start."""
    from visual_odometry_amd import geometry
    X, uv, R, t, bad = planar_problem(seed, n, outl)
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
    ok, rvec, tvec, inl = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
    assert ok and rc == 0
    assert np.array_equal(inl.ravel(), np.nonzero(mask)[0])
    assert np.abs(rvec.ravel() - rv).max() < POSE_TOL and np.abs(tvec.ravel() - tv).max() < POSE_TOL
    assert np.abs(geometry.Rodrigues(rvec)[0] - R).max() < 0.03 and np.abs(tvec.ravel() - t).max() < 0.15


def test_fast_mode_equals_its_oracle_and_agrees_with_cv2_mode(oracle, ctx, refine_modes):
    """vo_set_pnp_refine(0): same inlier sets; the pose is the same minimum reached from the RANSAC model (1e-9 against the
    oracle's fast mode, 1e-6 against cv2's mode — which stops at a FLT_EPSILON step — wherever cv2's mode refines at all:
    with exactly 5 non-planar inliers cv2 returns the RANSAC model itself, "DLT algorithm needs at least 6 points")."""
    from visual_odometry_amd import geometry
    five = 0
    for seed in range(100, 170):
        n = [8, 20, 100, 400][seed % 4]; outl = [0, 0.2, 0.5][seed % 3]
        X, uv, *_ = problem(seed, n, outl)
        refine_modes("cv2")
        ok1, r1, t1, i1 = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
        rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(X, uv, K)
        refine_modes("fast")
        ok0, r0, t0, i0 = geometry.solvePnPRansac(X, uv, K, np.zeros(4))
        rc0, rv0, tv0, mask0, _ = oracle.solve_pnp_ransac(X, uv, K)
        assert ok0 == ok1 == (rc == 0) == (rc0 == 0)
        if not ok0:
            continue
        assert np.array_equal(i0, i1) and np.array_equal(mask, mask0)
        assert np.abs(r0.ravel() - rv0).max() < 1e-9 and np.abs(t0.ravel() - tv0).max() < 1e-9
        assert np.abs(r1.ravel() - rv).max() < POSE_TOL and np.abs(t1.ravel() - tv).max() < POSE_TOL
        if ninl == 5:
            five += 1                                                   # cv2 mode: the un-refined RANSAC model
            continue
        assert np.abs(r0 - r1).max() < 1e-6 and np.abs(t0 - t1).max() < 1e-6
    assert five >= 1
