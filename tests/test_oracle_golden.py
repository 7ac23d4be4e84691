"""CPU: the oracle against the committed vectors and the only numeric known-answer the reference holds.

PARITY UNPINNED (see oracle/voo.h): tests/golden/*.npz were produced by the oracle itself
(tests/scripts/make_golden.py); they freeze its behaviour, they do not prove agreement with cv2."""
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_camera_matrix_known_answer():
    """/root/reference/test.g2o:1 holds the g2o CameraParameters the reference saved: focal 802.832 and
    principal point (565.427, 240.124) = set_camera_matrix (visual_slam.py:28-37)."""
    from visual_odometry_amd.synth import reference_camera_matrix
    k = reference_camera_matrix()
    assert abs(k[0, 0] - 802.832) < 5e-4 and abs(k[1, 1] - 802.832) < 5e-4
    assert abs(k[0, 2] - 565.427) < 5e-4 and abs(k[1, 2] - 240.124) < 5e-4
    assert k[2, 2] == 1 and k[0, 1] == 0


def test_pattern_table_checksum():
    import zlib
    src = open(os.path.join(os.path.dirname(G), "..", "oracle", "orb_pattern.inc")).read()
    body = src[src.index("*/") + 2:]
    vals = np.array([int(v) for v in body.replace("\n", " ").split(",") if v.strip()], np.int8)
    assert vals.shape == (1024,)
    assert vals[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12] and vals[-4:].tolist() == [-1, -6, 0, -11]
    assert zlib.crc32(vals.tobytes()) == 0xD1A39030
    assert np.abs(vals).max() == 13
    prod = open(os.path.join(os.path.dirname(G), "..", "visual_odometry_amd", "csrc", "orb_pattern.inc")).read()
    assert prod == src              # the product's table is the same published constant


def test_level_geometry_matches_survey(oracle):
    lw, lh, ls, q = oracle.level_geometry(720, 1280, oracle.orb_params(nfeatures=2000))
    assert lw.tolist() == [1280, 1067, 889, 741, 617, 514, 429, 357]
    assert lh.tolist() == [720, 600, 500, 417, 347, 289, 241, 201]
    assert q.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    _, _, _, q5 = oracle.level_geometry(480, 640, oracle.orb_params(nfeatures=500))
    assert q5.tolist() == [109, 90, 75, 63, 52, 44, 36, 31]
    lw4, lh4, _, q4 = oracle.level_geometry(1080, 1920, oracle.orb_params(nfeatures=4000, nlevels=4))
    assert q4.tolist() == [1288, 1073, 894, 745] and int((lw4.astype(np.int64) * lh4).sum()) == 5207725


def test_pair_golden(oracle):
    g = np.load(os.path.join(G, "pair_320x240.npz"))
    p = oracle.orb_params(nfeatures=int(g["nfeatures"]), nlevels=int(g["nlevels"]))
    f = g["frames"]
    lv = oracle.pyramid(f[0], p)
    assert np.array_equal(lv[1], g["level1"])
    assert np.array_equal(oracle.fast_score_nms(lv[0], 20), g["fast0"])
    assert np.array_equal(oracle.gaussian_blur7(lv[0]), g["blur0"])
    d0, d1 = oracle.orb_detect_and_compute(f[0], p), oracle.orb_detect_and_compute(f[1], p)
    for k in ("xy", "angle", "response", "octave", "desc"):
        assert np.array_equal(d0[k], g[k + "0"]), k
    assert np.array_equal(d1["xy"], g["xy1"]) and np.array_equal(d1["desc"], g["desc1"])
    q, t, d = oracle.match_hamming(d0["desc"], d1["desc"], 2)
    assert np.array_equal(q, g["cc_q"]) and np.array_equal(t, g["cc_t"]) and np.array_equal(d, g["cc_d"])
    q, t, d = oracle.match_hamming(d0["desc"], d1["desc"], 1)
    assert np.array_equal(q, g["legacy_q"]) and np.array_equal(t, g["legacy_t"]) and np.array_equal(d, g["legacy_d"])
    assert set(zip(g["cc_q"].tolist(), g["cc_t"].tolist())) <= set(zip(q.tolist(), t.tolist()))   # mutual pairs survive both rules
    q, t, d = oracle.knn2_ratio_hamming(d0["desc"], d1["desc"], 0.8)
    assert np.array_equal(q, g["ratio_q"]) and np.array_equal(t, g["ratio_t"]) and np.array_equal(d, g["ratio_d"])
    pr = oracle.pair(f[0], f[1], p, g["K"])
    assert (pr["n_match"], pr["n_inl"], pr["n_good"]) == (int(g["n_match"]), int(g["n_inl"]), int(g["n_good"]))
    assert np.allclose(pr["R"], g["R"], atol=1e-12) and np.allclose(pr["t"], g["t"], atol=1e-12)
    assert np.allclose(pr["X"], g["X"], rtol=1e-9)


def test_geometry_golden(oracle):
    g = np.load(os.path.join(G, "geometry_400.npz"))
    rc, E, mask, ninl = oracle.find_essential_ransac(g["p1"], g["p2"], g["K"])
    assert rc == 0 and ninl == int(g["n_inl"]) and np.array_equal(mask, g["mask"])
    assert np.allclose(E[0], g["E"], atol=1e-12)
    ng, R, t, pm = oracle.recover_pose(E[0], g["p1"][mask > 0], g["p2"][mask > 0], g["K"])
    assert ng == int(g["n_good"]) and np.array_equal(pm, g["pose_mask"])
    assert np.allclose(R, g["R"], atol=1e-12) and np.allclose(t, g["t"], atol=1e-12)
    # and it is a sane pose: close to the generating motion
    assert np.linalg.norm(R - g["R_true"]) < 0.05 and np.linalg.norm(t.ravel() - g["t_true"]) < 0.1


def test_ingest_golden(oracle):
    g = np.load(os.path.join(G, "ingest_192x108.npz"))
    for name, (dw, dh) in (("dst_57x32", (57, 32)), ("dst_96x54", (96, 54)), ("dst_250x120", (250, 120))):
        assert np.array_equal(oracle.resize_linear(g["src"], dw, dh), g[name])
    for name, (dw, dh) in (("area_96x54", (96, 54)), ("area_64x36", (64, 36)), ("area_57x32", (57, 32))):     # INTER_AREA
        assert np.array_equal(oracle.resize_area(g["src"], dw, dh), g[name])


def test_pnp_golden(oracle):
    g = np.load(os.path.join(G, "pnp_240.npz"))
    rc, rv, tv, mask, ninl = oracle.solve_pnp_ransac(g["obj"], g["img"], g["K"])
    assert rc == int(g["rc"]) == 0 and ninl == int(g["n_inl"]) and np.array_equal(mask, g["mask"])
    assert np.allclose(rv, g["rvec"], atol=1e-12) and np.allclose(tv, g["tvec"], atol=1e-12)
    assert np.abs(oracle.rodrigues(rv) - g["R_true"]).max() < 0.01 and np.abs(tv - g["t_true"]).max() < 0.05
