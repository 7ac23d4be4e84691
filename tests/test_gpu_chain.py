"""The step after the pair path on resident data: vo_tracks_pnp_batch (FrontEnd.localize_chain) against the three existing
oracles composed the way the reference composes the cv2 calls — /root/reference/src/visual_slam.py:183-266 (update_feature_mapper,
track_feature_back_in_time, estimate_current_camera_position) and :153-180 (add_information_to_map), the dict walks re-typed
below, cv2.solvePnPRansac / cv2.Rodrigues / cv2.triangulatePoints replaced by oracle.solve_pnp_ransac / rodrigues / triangulate.
No bundle adjustment on either side (src/map.py:104-186 is out of scope); the initial cameras are stored consistently with the
initial points (documented deviation, include/vo_hip.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _KP(K, T):
    """K @ T with the operation order of the kernels (k_triangulate_pairs, k_chain_pose): left-to-right sums of products, one rounding
    per operation — numpy's matmul may fuse or reorder them, and the chain amplifies last-bit differences of its inputs."""
    return np.array([[K[r, 0] * T[0, c] + K[r, 1] * T[1, c] + K[r, 2] * T[2, c] for c in range(4)] for r in range(3)])


def _inv_pose(R, t):
    """[R^T | -R^T t] as k_triangulate_pairs / k_chain_init form it."""
    t = np.asarray(t).ravel()
    return np.array([[R[0, r], R[1, r], R[2, r], -(R[0, r] * t[0] + R[1, r] * t[1] + R[2, r] * t[2])] for r in range(3)])


def reference_chain(O, feats, pairs, K, max_norm=50.0, follow=None):
    """The reference's loop over a sequence, on the oracle's stage outputs.  feats[f] = dict(xy, desc).
    follow = the device's poses [n, 3, 4]: every solvePnPRansac result is reported as computed, but the chain then continues from
    the DEVICE's camera, so that every step is compared on identical inputs (a RANSAC over points near its 8-pixel threshold turns
    a 1e-9 difference of the previous pose into a different winner: the free-running chains agree only while the geometry is good)."""
    feature_mapper, mappoints, cameras = {}, {}, {}
    out = dict(poses=[], n_corr=[], n_inl=[], status=[], n_map=[], E_inl=[], corr=[None])
    alive = True
    for p, (a, b) in enumerate(pairs):
        qi, ti, _ = O.match_hamming(feats[a]["desc"], feats[b]["desc"], 2)
        p1 = feats[a]["xy"][qi].astype(np.float64); p2 = feats[b]["xy"][ti].astype(np.float64)
        rc, E, mask, _ = O.find_essential_ransac(p1, p2, K)
        assert rc == 0
        inl = mask > 0
        _, R, t, _ = O.recover_pose(E[0], p1[inl], p2[inl], K)
        m3d = [((a, int(q)), (b, int(tt)), kp2) for q, tt, kp2 in zip(qi[inl], ti[inl], p2[inl])]     # matches_with_3d_information
        out["E_inl"].append(int(inl.sum()))
        for fid1, fid2, _ in m3d:                                      # update_feature_mapper (:183-188) / initialize_map (:48-52)
            feature_mapper[fid2] = fid1
        if p == 0:                                                     # initialize_map (:43-92), cameras consistent with the points
            X = O.triangulate(_KP(K, _inv_pose(R, t)), _KP(K, np.eye(3, 4)), p1[inl].T, p2[inl].T)
            X = X / X[3]
            cameras[a] = _inv_pose(R, t); cameras[b] = np.eye(3, 4)
            for idx, (fid1, _, _) in enumerate(m3d):
                mappoints[fid1] = X[:3, idx].copy()
            out["poses"] += [cameras[a], cameras[b]]
            out["n_corr"].append(0); out["n_inl"].append(0); out["status"].append(0); out["n_map"].append(len(mappoints))
            continue
        if not alive:
            out["corr"].append(None)
            out["poses"].append(np.zeros((3, 4))); out["n_corr"].append(0); out["n_inl"].append(0); out["status"].append(None); out["n_map"].append(len(mappoints))
            continue
        obj, img = [], []                                              # estimate_current_camera_position (:201-227)
        for fid1, fid2, kp2 in m3d:
            feature_id = fid2
            while feature_id in feature_mapper:
                feature_id = feature_mapper[feature_id]
            if feature_id in mappoints:
                obj.append(mappoints[feature_id]); img.append(kp2)
        out["n_corr"].append(len(obj)); out["corr"].append((np.array(obj).reshape(-1, 3), np.array(img).reshape(-1, 2)))
        rc, rvec, tvec, pmask, ninl = O.solve_pnp_ransac(np.array(obj).reshape(-1, 3), np.array(img).reshape(-1, 2), K) if len(obj) >= 4 else (-1, None, None, None, 0)
        out["n_inl"].append(int(ninl))
        if rc != 0:                                                    # cv2 raises / retval False: no camera is added (:253-261)
            alive = False
            out["poses"].append(np.zeros((3, 4))); out["status"].append(rc); out["n_map"].append(len(mappoints))
            continue
        out["status"].append(0)
        cameras[b] = np.hstack([O.rodrigues(rvec), tvec.reshape(3, 1)])   # R, _ = cv2.Rodrigues(rvec); TrackedCamera(R, tvec) (:243-249)
        out["poses"].append(cameras[b])
        if follow is not None:
            cameras[b] = np.array(follow[p + 1], np.float64)
        X = O.triangulate(_KP(K, cameras[a]), _KP(K, cameras[b]), p1[inl].T, p2[inl].T)     # add_information_to_map (:164-172)
        X = X / X[3]
        snapshot = dict(mappoints)                                     # self.mappointdict is built before the loop (:154-156)
        for idx, (fid1, fid2, _) in enumerate(m3d):
            if not np.linalg.norm(X[:3, idx]) <= max_norm:
                continue
            feature_id = fid2
            while feature_id in feature_mapper:
                feature_id = feature_mapper[feature_id]
            if feature_id in snapshot:
                continue                                               # add_new_observation_of_existing_point: nothing to store without BA
            mappoints[fid1] = X[:3, idx].copy()                        # add_new_match_to_map: keyed by featureid1
        out["n_map"].append(len(mappoints))
    return out


def _advances_along_a_line(poses):
    centres = np.array([-P[:, :3].T @ P[:, 3] for P in poses])
    d = np.diff(centres, axis=0)
    steps = np.linalg.norm(d, axis=1)
    assert np.all((steps > 0.6) & (steps < 1.4)), steps
    assert np.all((d @ d[0]) / (steps * steps[0]) > 0.9), (steps, d)


@pytest.mark.parametrize("w,h,n,nfeat,physical", [(640, 480, 7, 1000, False), (1280, 720, 5, 2000, True)])
def test_chain_equals_the_composed_oracles(oracle, kernel_dk_rule, w, h, n, nfeat, physical):
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(n, w, h, step=4.0, cache_dir="/tmp")             # 4 units per frame, 30 above the ground: enough parallax for a BA-free chain
    K = seq["K"]
    pairs = [[k, k + 1] for k in range(n - 1)]
    fe = FrontEnd(h, w, max_frames=n, max_pairs=n - 1, nfeatures=nfeat)
    fe.upload(seq["frames"]); fe.detect(0, n)
    res, _ = fe.run_pairs(pairs, K, want_points=True)
    res = res.copy()
    got = fe.localize_chain(n - 1, K)
    p = oracle.orb_params(nfeatures=nfeat)
    feats = [oracle.orb_detect_and_compute(seq["frames"][f], p) for f in range(n)]
    want = reference_chain(oracle, feats, pairs, K)
    assert [int(v) for v in res["n_inl"]] == want["E_inl"]
    assert got["status"].tolist() == want["status"] == [0] * (n - 1)
    assert got["n_corr"].tolist() == want["n_corr"] and got["n_inl"].tolist() == want["n_inl"]
    assert got["n_map"].tolist() == want["n_map"]
    assert min(want["n_corr"][1:]) > 50 and min(want["n_inl"][1:]) > 30
    for k in range(n):
        assert np.abs(got["poses"][k] - want["poses"][k]).max() < 1e-6, k
    # what the chain is for: the baselines of consecutive frames in ONE scale (the E chain alone gives unit steps).  The flight
    # moves the same distance per frame along a nearly straight line and pair 0's baseline is the unit, so every later step is about
    # one unit long too.  (Not asserted at 640x480 / 1000 features: there solvePnPRansac — EPnP hypotheses on nearly planar ground —
    # lands on a far-off pose at the second pair, in the oracle chain exactly as here; the reference relies on g2o after every frame.)
    if physical:
        _advances_along_a_line(got["poses"])


def test_chain_rejects_what_it_cannot_walk(oracle):
    from visual_odometry_amd import _lib, synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(4, 640, 480, cache_dir="/tmp")
    fe = FrontEnd(480, 640, max_frames=4, max_pairs=3, nfeatures=500)
    fe.upload(seq["frames"]); fe.detect(0, 4)
    fe.run_pairs([[0, 1], [1, 2]], seq["K"], want_points=False)
    with pytest.raises(_lib.VoError):                                    # no triangulated points in HBM
        fe.localize_chain(2, seq["K"])
    fe.run_pairs([[0, 1], [2, 3]], seq["K"], want_points=True)
    with pytest.raises(_lib.VoError):                                    # not a chain
        fe.localize_chain(2, seq["K"])
    fe.run_pairs([[0, 1], [1, 2], [2, 3]], seq["K"], want_points=True)
    with pytest.raises(_lib.VoError):                                    # all pairs of the run, not a part of them
        fe.localize_chain(2, seq["K"])
    out = fe.localize_chain(3, seq["K"])
    assert out["status"].tolist() == [0, 0, 0]
    # a chain that cannot continue: frame 2 is blank (no keypoints -> pair (1, 2) fails in vo_pairs_run, the chain stops there)
    frames = seq["frames"].copy(); frames[2] = 127
    fe.upload(frames); fe.detect(0, 4)
    res, _ = fe.run_pairs([[0, 1], [1, 2], [2, 3]], seq["K"], want_points=True)
    assert res["status"][1] != 0
    out = fe.localize_chain(3, seq["K"])
    assert out["status"][0] == 0 and out["status"][1] == res["status"][1] and out["status"][2] == _lib.VO_ERR_NOT_CONFIGURED
    assert np.all(out["poses"][2:] == 0)


def test_sift_chain_runs_on_the_live_configuration(oracle):
    """The same chain behind the batched SIFT + L2 front end (the reference's live configuration feeds exactly this step)."""
    from visual_odometry_amd import synth
    from visual_odometry_amd.frontend import FrontEnd
    n = 5
    seq = synth.sequence(n, 640, 360, step=4.0, cache_dir="/tmp")
    fe = FrontEnd(360, 640, max_frames=n, max_pairs=n - 1, detector="sift", kp_cap=4096)
    fe.upload(seq["frames"]); fe.detect(0, n)
    fe.run_pairs([[k, k + 1] for k in range(n - 1)], seq["K"], want_points=True)
    out = fe.localize_chain(n - 1, seq["K"])
    assert out["status"].tolist() == [0] * (n - 1) and out["n_inl"][1:].min() > 30
    _advances_along_a_line(out["poses"])
