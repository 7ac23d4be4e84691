"""GPU parity of the first widened row: the map's reprojection-error filter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Cam:
    def __init__(self, cid, R, t): self.camera_id, self.R, self.t = cid, R, t
    def pose(self):
        T = np.eye(4); T[:3, :3] = self.R; T[:3, 3] = np.asarray(self.t).ravel(); return T


class Pt:
    def __init__(self, pid, p): self.point_id, self.point = pid, tuple(p)


class Obs:
    def __init__(self, pid, cid, xy): self.point_id, self.camera_id, self.image_coordinates = pid, cid, xy


def _map(seed, ncam, npt, nobs):
    rng = np.random.default_rng(seed)
    K = np.array([[802.8, 0, 565.4], [0, 802.8, 240.1], [0, 0, 1.0]])
    cams = []
    for i in range(ncam):
        a = rng.normal(0, 0.05, 3); th = np.linalg.norm(a); k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        cams.append(Cam(100 + 7 * i, np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx, rng.normal(0, 0.5, 3)))
    pts = [Pt(5000 + 3 * j, rng.uniform(-5, 5, 3) + [0, 0, 20]) for j in range(npt)]
    obs = []
    for _ in range(nobs):
        c, p = cams[rng.integers(ncam)], pts[rng.integers(npt)]
        x = K @ (c.pose() @ np.r_[p.point, 1.0])[:3]
        obs.append(Obs(p.point_id, c.camera_id, tuple(x[:2] / x[2] + rng.normal(0, rng.choice([0.5, 3, 12]), 2))))
    return K, cams, pts, obs


def test_reprojection_filter_matches_oracle_and_numpy(oracle, ctx):
    from visual_odometry_amd import map_filters as mf
    K, cams, pts, obs = _map(1, 18, 3000, 20000)
    arrays = mf._arrays(cams, pts, obs)
    err, keep = mf.reprojection_sqerr(*arrays, K, 100)
    eo, ko = oracle.reprojection_sqerr(*arrays, K, 100)
    assert np.array_equal(err, eo) and np.array_equal(keep, ko)          # same left-to-right float64 arithmetic
    # numpy, as the reference writes it (map.py:57-64)
    for i in range(0, 20000, 997):
        o = obs[i]
        cam = next(c for c in cams if c.camera_id == o.camera_id); p = next(p for p in pts if p.point_id == o.point_id)
        t = K @ (cam.pose() @ np.array([np.hstack((np.array(p.point), 1))]).T)[0:3, :]
        t = t / t[2, 0]
        sq = float(np.abs((t[0, 0] - o.image_coordinates[0]) ** 2) + np.abs((t[1, 0] - o.image_coordinates[1]) ** 2))
        assert abs(sq - err[i]) <= 1e-9 * max(1.0, sq)
    kept = mf.remove_observations_with_reprojection_errors_above_threshold(cams, pts, obs, K, 100)
    assert len(kept) == int(keep.sum()) and 0.4 * len(obs) < len(kept) < len(obs)
    total = mf.calculate_reprojection_error(cams, pts, obs, K)
    assert abs(total - float(np.sum(eo))) <= 1e-9 * total


def test_reprojection_filter_edge_cases(ctx):
    from visual_odometry_amd import map_filters as mf, _lib
    K = np.eye(3)
    assert mf.remove_observations_with_reprojection_errors_above_threshold([], [], [], K) == []
    with pytest.raises(_lib.VoError):
        mf.reprojection_sqerr(np.eye(4)[None], np.zeros((1, 3)), [0], [5], [[0, 0]], K)     # missing point


def test_feature_tracks_match_the_reference_dict_walk(ctx, seq_small):
    """feature_mapper as the reference builds it (a dict keyed by (frame, idx)) against the array version, on real
    inlier matches of consecutive pairs plus a skip pair that overwrites entries (last assignment wins)."""
    from visual_odometry_amd import map_filters as mf
    from visual_odometry_amd.frontend import FrontEnd
    frames, K = seq_small["frames"], seq_small["K"]
    fe = FrontEnd(480, 640, max_frames=4, max_pairs=4, nfeatures=500)
    fe.upload(frames); fe.detect(0, 4)
    pairs = [[0, 1], [1, 2], [0, 2], [2, 3]]
    fe.run_pairs(pairs, K)
    matches = []
    for p in range(len(pairs)):
        qi, ti, _, mask = fe.pair_matches(p)
        matches.append((qi[mask > 0], ti[mask > 0]))
    cap = fe.kp_cap
    rf, ri, hops = mf.feature_tracks(4, cap, pairs, matches)
    mapper = {}
    for (f1, f2), (q, t) in zip(pairs, matches):          # update_feature_mapper, pair after pair
        for a, b in zip(q.tolist(), t.tolist()):
            mapper[(f2, b)] = (f1, a)
    longest = 0
    for f in range(4):
        for i in range(0, cap, 3):
            fid, n = (f, i), 0
            while fid in mapper:                          # track_feature_back_in_time
                fid = mapper[fid]; n += 1
            assert (rf[f, i], ri[f, i], hops[f, i]) == (fid[0], fid[1], n)
            longest = max(longest, n)
    assert longest >= 2
    with pytest.raises(Exception):
        mf.feature_tracks(2, 8, [[0, 1], [1, 0]], [(np.array([0]), np.array([0])), (np.array([0]), np.array([0]))])   # a cycle


@pytest.mark.parametrize("shape,dsize", [((2160, 3840, 3), (1152, 648)), ((216, 384, 3), (115, 64)), ((97, 131), (64, 48)),
                                         ((50, 70, 4), (140, 100)), ((60, 80, 3), (40, 30)), ((33, 47), (200, 9)),
                                         ((5, 5, 3), (1, 1)), ((1, 1), (7, 3))])
def test_ingest_resize_bit_exact(oracle, ctx, shape, dsize):
    """cv2.resize(img, dim) (visual_slam.py:346-352; the reference's 3840x2160 -> 1152x648 case first)."""
    from visual_odometry_amd import ingest
    rng = np.random.default_rng(sum(shape) + dsize[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(ingest.resize(img, dsize), oracle.resize_linear(img, dsize[0], dsize[1]))


def test_ingest_into_the_front_end(oracle, ctx):
    """Full-resolution BGR frames -> device resize -> gray -> ORB must equal oracle resize + oracle ORB on the result."""
    from visual_odometry_amd import ingest, synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(2, 1280, 720, cache_dir="/tmp")
    big = np.stack([np.stack([f, np.roll(f, 3, 1), 255 - f], axis=2) for f in seq["frames"]])       # [2, 720, 1280, 3]
    fe = FrontEnd(216, 384, max_frames=2, max_pairs=1, nfeatures=500)
    small = fe.ingest(big, want_resized=True)
    for i in range(2):
        assert np.array_equal(small[i], oracle.resize_linear(big[i], 384, 216))
    fe.detect(0, 2)
    ref = oracle.orb_detect_and_compute(oracle.resize_linear(big[1], 384, 216), oracle.orb_params(nfeatures=500))
    got = fe.features(1)
    assert np.array_equal(got["xy"], ref["xy"]) and np.array_equal(got["desc"], ref["desc"])
    gray_small = fe.ingest(seq["frames"])                                         # gray input goes straight to level 0
    assert gray_small is None
    fe.detect(0, 1)
    ref = oracle.orb_detect_and_compute(oracle.resize_linear(seq["frames"][0], 384, 216), oracle.orb_params(nfeatures=500))
    assert np.array_equal(fe.features(0)["desc"], ref["desc"])
    with pytest.raises(NotImplementedError):
        ingest.resize(big[0], (2000, 1000), interpolation=ingest.INTER_AREA)     # enlarging INTER_AREA is not built


@pytest.mark.parametrize("shape,dsize", [((2160, 3840, 3), (1152, 648)), ((216, 384, 3), (192, 108)), ((216, 384, 3), (128, 72)),
                                         ((216, 384), (96, 108)), ((216, 384, 3), (115, 64)), ((97, 131), (64, 48)),
                                         ((60, 80, 4), (80, 60)), ((60, 80, 4), (79, 59)), ((33, 47), (1, 1)), ((5, 5, 3), (5, 2))])
def test_inter_area_resize_bit_exact(oracle, ctx, shape, dsize):
    """cv2.resize(img, dim, interpolation=cv2.INTER_AREA) (image_and_keypoints.py:42): integer factors (2 x 2 and other
    blocks), mixed integer factors, fractional factors (the reference's 0.3 among them), identity, 1, 3 and 4 channels."""
    from visual_odometry_amd import ingest
    rng = np.random.default_rng(sum(shape) + dsize[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = ingest.resize(img, dsize, interpolation=ingest.INTER_AREA)
    assert got.shape[:2] == (dsize[1], dsize[0]) and np.array_equal(got, oracle.resize_area(img, dsize[0], dsize[1]))


def test_image_and_keypoints_rescales_with_inter_area(oracle, ctx, seq_small):
    from visual_odometry_amd import ImageAndKeypoints
    iak = ImageAndKeypoints("ORB")
    iak.scale_factor = 0.5
    bgr = np.stack([seq_small["frames"][0]] * 3, axis=2)
    iak.set_image(bgr)
    assert iak.image.shape == (240, 320, 3) and np.array_equal(iak.image, oracle.resize_area(bgr, 320, 240))
    iak.detect_keypoints()
    assert len(iak.keypoints) > 50
