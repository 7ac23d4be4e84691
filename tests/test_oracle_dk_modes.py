"""CPU: the oracle's two root-finder rules inside the five-point solver.

cv::solvePoly runs a fixed 300 Durand-Kerner sweeps; that is the oracle's default.  The HIP kernel's throughput
mode stops a sample once further sweeps only move rounding noise (oracle.set_dk_early_exit(True) restates exactly
that rule, including its 64-sweep cap).  These tests pin what the shortcut may change: nothing integer (number and order of models, RANSAC
decisions, inlier masks), and floats far below the north star's tolerances ([R|t] 1e-4, points 1e-3)."""
import os

import numpy as np
import pytest
from twoview import five_point_sample, fuzz_problem, scene

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture
def both(oracle):
    def run(fn):
        assert not oracle.get_dk_early_exit()             # the default is the faithful 300-sweep form
        a = fn()
        oracle.set_dk_early_exit(True)
        try:
            b = fn()
        finally:
            oracle.set_dk_early_exit(False)
        return a, b
    return run


def test_five_point_models_agree(oracle, both):
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(200):
        x1, x2 = five_point_sample(rng)
        full, early = both(lambda: oracle.five_point(x1, x2))
        assert full.shape == early.shape                  # same real roots, same order
        if len(full):
            worst = max(worst, float(np.abs(full - early).max()))
    # near-double roots sit at their conditioning floor either way; the rare sample that is still creeping towards its
    # roots when the throughput rule gives up (64 sweeps) differs by ~1e-5, an order below the north star's 1e-4
    assert worst < 1e-4, worst


def _solve(oracle, K, p1, p2, **kw):
    rc, E, mask, ninl = oracle.find_essential_ransac(p1, p2, K, **kw)
    if rc != 0:
        return rc, None, mask, None, None
    inl = mask > 0
    ng, R, t, pm = oracle.recover_pose(E[0], p1[inl], p2[inl], K)
    return rc, E[0], mask, np.hstack([R, t]), pm


def test_golden_geometry_same_decisions(oracle, both):
    g = np.load(os.path.join(G, "geometry_400.npz"))
    full, early = both(lambda: _solve(oracle, g["K"], g["p1"], g["p2"]))
    assert full[0] == early[0] == 0 and np.array_equal(full[2], early[2]) and np.array_equal(full[4], early[4])
    assert np.linalg.norm(full[3] - early[3]) < 1e-9


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_same_masks_and_pose(oracle, both, seed):
    """The 30 problems of tests/test_gpu_geometry.py::test_randomised_two_view_problems plus five denser scenes."""
    rng = np.random.default_rng(500 + seed)
    problems = [fuzz_problem(rng) for _ in range(5)]
    K, _, _, p1, p2 = scene(100 + seed, 600, outliers=0.4)
    problems.append(dict(K=K, p1=p1, p2=p2, thresh=1.0, prob=0.99, tag="scene"))
    for pr in problems:
        full, early = both(lambda: _solve(oracle, pr["K"], pr["p1"], pr["p2"], prob=pr["prob"], thresh=pr["thresh"]))
        assert full[0] == early[0], pr["tag"]
        if full[0] != 0:
            continue
        assert np.array_equal(full[2], early[2]), pr["tag"]          # identical inlier masks
        assert np.array_equal(full[4], early[4]), pr["tag"]          # identical cheirality masks
        assert np.linalg.norm(full[3] - early[3]) < 1e-4, pr["tag"]   # north star tolerance on [R|t] ...
        assert np.linalg.norm(full[1] - early[1]) < 1e-7, pr["tag"]   # ... and in fact the same model to rounding noise
