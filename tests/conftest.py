import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree libraries are git-ignored build products: always run make (a no-op when they are up to date), so a
    # stale binary is never what gets tested.  The GPU box has the same toolchain; without one, use what travelled.
    import shutil
    have = os.path.exists(os.path.join(ROOT, "visual_odometry_amd", "libvo_hip.so")) and \
        os.path.exists(os.path.join(ROOT, "oracle", "libvoo.so"))
    if shutil.which("hipcc") and shutil.which("make") or not have:
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture
def kernel_dk_rule(oracle):
    """The oracle's five-point root finder switched to the HIP kernel's throughput rule (Durand-Kerner sweeps stop at
    the rounding-noise floor) for the duration of a test, so that E / masks / [R|t] can be compared bit for bit with
    the product's default mode.  Without it the oracle runs cv::solvePoly's fixed 300 sweeps."""
    oracle.set_dk_early_exit(True)
    yield oracle
    oracle.set_dk_early_exit(False)


@pytest.fixture(scope="session")
def seq_small():
    """4 frames 640x480 of the seeded synthetic drone sequence (BASELINE config 1 shape)."""
    from visual_odometry_amd import synth
    return synth.sequence(4, 640, 480, cache_dir="/tmp")


@pytest.fixture(scope="session")
def ctx():
    from visual_odometry_amd import _lib
    return _lib.default_context(0)


def random_image(seed, h, w):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1)).astype(np.float32)
    img = np.kron(img, np.ones((4, 4), np.float32))[:h, :w]
    img += rng.normal(0, 12, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)
