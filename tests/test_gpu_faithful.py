"""GPU parity against the FAITHFUL oracle (cv::solvePoly's fixed 300 Durand-Kerner sweeps, the oracle's default) and
BASELINE config 2 at its own size.

 * HIP in its `opencv300` mode (vo_set_poly_solver(ctx, 1)) against the faithful oracle: bit for bit.
 * HIP in its default throughput mode against the faithful oracle: identical integer results (counts, match pairs,
   inlier masks), [R|t] within 1e-4 (Frobenius), points within 1e-3 relative — the north star's tolerances.
 * Eight consecutive pairs of the 1280x720 / 2000-feature / 8-level synthetic sequence through the batched front end,
   every per-pair output compared (counts, match index pairs, mask, E, R|t, X)."""
import numpy as np
import pytest
from twoview import five_point_sample, fuzz_problem, oracle_pair_stages

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx300(ctx, oracle):
    assert not oracle.get_dk_early_exit()
    ctx.set_poly_solver("opencv300")
    yield ctx
    ctx.set_poly_solver("fast")


def test_five_point_300_sweeps_bit_identical(oracle, ctx300):
    from visual_odometry_amd import geometry
    rng = np.random.default_rng(0)
    for _ in range(40):
        x1, x2 = five_point_sample(rng)
        assert np.array_equal(geometry.five_point(x1, x2), oracle.five_point(x1, x2))


@pytest.mark.parametrize("seed", range(3))
def test_two_view_fuzz_300_sweeps_bit_identical(oracle, ctx300, seed):
    from visual_odometry_amd import geometry
    rng = np.random.default_rng(500 + seed)
    for _ in range(5):
        pr = fuzz_problem(rng)
        rc, Es, mask, ninl = oracle.find_essential_ransac(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], thresh=pr["thresh"])
        E, m = geometry.findEssentialMat(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], threshold=pr["thresh"])
        if rc != 0:
            assert E is None, pr["tag"]
            continue
        assert np.array_equal(m.ravel(), mask) and np.array_equal(E, Es[0]), pr["tag"]
        inl = mask > 0
        ng, Rr, tr, pm = oracle.recover_pose(Es[0], pr["p1"][inl], pr["p2"][inl], pr["K"])
        ng2, R2, t2, pm2 = geometry.recoverPose(E, pr["p1"][inl], pr["p2"][inl], pr["K"])
        assert ng2 == ng and np.array_equal(R2, Rr) and np.array_equal(t2, tr), pr["tag"]


@pytest.mark.parametrize("seed", range(6))
def test_two_view_fuzz_default_mode_vs_faithful_oracle(oracle, ctx, seed):
    """The product's default root-finder rule against OpenCV's: same decisions, floats within the north star."""
    from visual_odometry_amd import geometry
    assert not oracle.get_dk_early_exit()
    rng = np.random.default_rng(500 + seed)
    for _ in range(5):
        pr = fuzz_problem(rng)
        rc, Es, mask, ninl = oracle.find_essential_ransac(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], thresh=pr["thresh"])
        E, m = geometry.findEssentialMat(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], threshold=pr["thresh"])
        if rc != 0:
            assert E is None, pr["tag"]
            continue
        assert np.array_equal(m.ravel(), mask), pr["tag"]
        inl = mask > 0
        ng, Rr, tr, pm = oracle.recover_pose(Es[0], pr["p1"][inl], pr["p2"][inl], pr["K"])
        ng2, R2, t2, pm2 = geometry.recoverPose(E, pr["p1"][inl], pr["p2"][inl], pr["K"])
        assert ng2 == ng and np.array_equal(pm2.ravel() > 0, pm > 0), pr["tag"]
        assert np.linalg.norm(np.hstack([R2, t2]) - np.hstack([Rr, tr])) < 1e-4, pr["tag"]


@pytest.fixture(scope="module")
def config2():
    """BASELINE config 2 at its own size: 9 consecutive 1280x720 frames, 2000 features, 8 levels -> 8 pairs, detected
    and matched once on the GPU; the oracle's stages are computed once (faithful root finder)."""
    from oracle import oracle as O
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd
    n = 9
    seq = synth.sequence(n, 1280, 720, cache_dir="/tmp")
    frames, K = seq["frames"], seq["K"]
    fe = FrontEnd(720, 1280, max_frames=n, max_pairs=n - 1, nfeatures=2000, nlevels=8)
    fe.upload(frames); fe.detect(0, n)
    p = O.orb_params(nfeatures=2000, nlevels=8)
    assert not O.get_dk_early_exit()
    ref = [oracle_pair_stages(O, frames[i], frames[i + 1], p, K) for i in range(n - 1)]
    return dict(fe=fe, K=K, n=n, ref=ref, pairs=[[i, i + 1] for i in range(n - 1)])


def _compare(c, exact):
    fe, K, ref = c["fe"], c["K"], c["ref"]
    res, X = fe.run_pairs(c["pairs"], K, fe.make_opts(want_points=True))
    for k, r in enumerate(ref):
        g = res[k]
        assert g["status"] == 0 and r["rc"] == 0, k
        assert (g["n_kp1"], g["n_kp2"]) == (len(r["d1"]["xy"]), len(r["d2"]["xy"])), k
        if k == 0:                                            # keypoints and descriptors of both frames, bit for bit
            for slot, d in ((0, r["d1"]), (1, r["d2"])):
                got = fe.features(slot)
                assert not got["truncated"]
                for key in ("xy", "octave", "angle", "response", "size", "desc"):
                    assert np.array_equal(got[key], d[key]), (slot, key)
        qi, ti, dist, mask = fe.pair_matches(k)
        assert np.array_equal(qi, r["qi"]) and np.array_equal(ti, r["ti"]) and np.array_equal(dist, r["dist"]), k   # match index pairs
        assert np.array_equal(mask, r["mask"]), k                                                                  # E-RANSAC inlier mask
        assert (g["n_match"], g["n_inl"], g["n_good"]) == (len(r["qi"]), r["n_inl"], r["n_good"]), k
        got_rt = np.hstack([g["R"].reshape(3, 3), g["t"].reshape(3, 1)])
        ref_rt = np.hstack([r["R"], r["t"]])
        n = r["n_inl"]
        if exact:
            assert np.array_equal(g["E"].reshape(3, 3), r["E"]) and np.array_equal(got_rt, ref_rt), k
            # the helper builds P = K [R^T | -R^T t] with numpy's matmul, whose summation order is not the kernel's
            nrm = np.linalg.norm(r["X"][:3], axis=0)
            near = nrm <= 10.0 * np.median(nrm)
            assert (np.linalg.norm(X[k][:3, :n] - r["X"][:3], axis=0) / nrm)[near].max() < 1e-9, k
        else:
            assert np.linalg.norm(got_rt - ref_rt) < 1e-4, k                                                       # north star: [R|t]
            sE = np.sign(np.sum(g["E"].reshape(3, 3) * r["E"]))
            assert np.linalg.norm(sE * g["E"].reshape(3, 3) - r["E"]) < 1e-4, k
            nrm = np.linalg.norm(r["X"][:3], axis=0)
            rel = np.linalg.norm(X[k][:3, :n] - r["X"][:3], axis=0) / nrm
            near = nrm <= 10.0 * np.median(nrm)             # points without parallax are arbitrarily ill conditioned
            assert near.sum() > 0.9 * n and rel[near].max() < 1e-3, (k, rel[near].max())                           # north star: points
    return res.copy()


def test_config2_pairs_default_mode_vs_faithful_oracle(config2):
    res = _compare(config2, exact=False)
    assert res["n_inl"].min() > 200 and res["n_match"].min() > 500          # a healthy, textured sequence


def test_config2_pairs_300_sweeps_bit_identical(config2):
    config2["fe"].ctx.set_poly_solver("opencv300")
    try:
        _compare(config2, exact=True)
    finally:
        config2["fe"].ctx.set_poly_solver("fast")
