"""CPU: the C-ABI library loads and exports every symbol include/vo_hip.h declares; host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from visual_odometry_amd import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vo_hip.h but not exported by libvo_hip.so"
    assert sorted(_lib.exported_symbols()) == names          # the ctypes table covers the whole header
    assert lib.vo_version() >= 100
    assert lib.vo_stage_name(2) == b"fast_score_nms"


def test_struct_layouts_match_the_header():
    from visual_odometry_amd import _lib
    assert ctypes.sizeof(_lib.OrbParams) == 36
    assert _lib.PAIR_RESULT_DTYPE.itemsize == 8 * 4 + 21 * 8
    assert ctypes.sizeof(_lib.PairOpts) == 64
    assert _lib.PAIR_RESULT_DTYPE.fields["R"][1] == 32 and _lib.PAIR_RESULT_DTYPE.fields["E"][1] == 32 + 96


def test_product_never_imports_the_oracle():
    for top in ("visual_odometry_amd", "examples", "tools", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".sh")):
                    text = open(os.path.join(dp, f)).read()
                    assert "libvoo" not in text and "from oracle" not in text and "import oracle" not in text, (top, f)


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from visual_odometry_amd import _lib
    with pytest.raises(_lib.VoError):
        _lib.Context(0)


def test_namedtuple_schemas_and_records():
    from visual_odometry_amd import Feature, Match, Match3D, Frame, KeyPoint, DMatch
    assert Feature._fields == ("keypoint", "descriptor", "feature_id")
    assert Match._fields == ("featureid1", "featureid2", "keypoint1", "keypoint2", "descriptor1", "descriptor2", "distance", "color")
    assert Match3D._fields == Match._fields + ("point",)
    f = Frame("img")
    assert (f.image, f.id, f.keypoints, f.descriptors, f.features) == ("img", None, None, None, None)
    f.id = 7
    assert repr(f) == repr("Frame 7")
    k = KeyPoint(1.5, 2, 31, 90, 0.1, 2)
    assert k.pt == (1.5, 2.0) and k.octave == 2
    m = DMatch(3, 4, 17)
    assert (m.queryIdx, m.trainIdx, m.distance) == (3, 4, 17.0)


def test_frame_generator_with_stub_detector():
    from visual_odometry_amd import FrameGenerator, KeyPoint

    class Stub:
        def detectAndCompute(self, image, mask):
            assert mask is None
            return (KeyPoint(1, 2), KeyPoint(3, 4)), np.arange(64, dtype=np.uint8).reshape(2, 32)
    g = FrameGenerator(Stub())
    a, b = g.make_frame("A"), g.make_frame("B")
    assert (a.id, b.id) == (0, 1)
    assert [f.feature_id for f in b.features] == [(1, 0), (1, 1)]
    assert b.features[1].keypoint.pt == (3.0, 4.0) and b.features[1].descriptor[0] == 32


def test_dropin_module_names_resolve():
    import importlib
    import sys
    d = os.path.join(ROOT, "visual_odometry_amd", "dropin")
    sys.path.insert(0, d)
    try:
        for name, attr in (("frame", "Frame"), ("frame_generator", "FrameGenerator"), ("initials", "Match3D")):
            sys.modules.pop(name, None)
            assert hasattr(importlib.import_module(name), attr)
    finally:
        sys.path.remove(d)
        for name in ("frame", "frame_generator", "initials", "image_pair"):
            sys.modules.pop(name, None)


def test_euler_helpers():
    from visual_odometry_amd.image_pair import isRotationMatrix, rotationMatrixToEulerAngles
    c, s = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    assert isRotationMatrix(R) and not isRotationMatrix(2 * R)
    assert np.allclose(rotationMatrixToEulerAngles(R), [0, 0, 0.3])


def test_synth_is_seeded_and_consistent():
    from visual_odometry_amd import synth
    a = synth.sequence(2, 160, 120)
    b = synth.sequence(2, 160, 120)
    assert np.array_equal(a["frames"], b["frames"]) and a["frames"].std() > 20
    R, t = synth.relative_pose(a["R"][0], a["C"][0], a["R"][1], a["C"][1])
    assert abs(np.linalg.norm(t) - 1) < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12


def test_chain_poses_inverts_steps():
    from visual_odometry_amd.frontend import chain_poses
    R = np.stack([np.eye(3)] * 3); t = np.tile(np.array([1.0, 0, 0]), (3, 1))
    T = chain_poses(R, t)
    assert T.shape == (4, 4, 4) and np.allclose(T[3][:3, 3], [-3, 0, 0])
