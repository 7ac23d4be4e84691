import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree libraries are git-ignored build products: build them if a fresh checkout lacks them
    if not (os.path.exists(os.path.join(ROOT, "visual_odometry_amd", "libvo_hip.so")) and
            os.path.exists(os.path.join(ROOT, "oracle", "libvoo.so"))):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


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
