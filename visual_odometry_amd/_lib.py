"""ctypes binding of libvo_hip.so (the C ABI declared in include/vo_hip.h).

The HIP library is the product: if it is missing or does not load, importing this module raises —
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VO_HIP_LIBRARY") or os.path.join(_HERE, "libvo_hip.so")   # VO_HIP_LIBRARY: an experimental build of the same ABI

VO_OK, VO_WARN_CAPACITY = 0, 1
VO_ERR_INVALID, VO_ERR_HIP, VO_ERR_TOO_FEW, VO_ERR_NO_MODEL, VO_ERR_NOT_CONFIGURED, VO_ERR_AMBIGUOUS = -1, -2, -3, -4, -5, -6
VO_ERR_UNSUPPORTED = -7
VO_STAGE_COUNT = 24
VO_COMM_ID_BYTES, VO_RECORD_DOUBLES = 128, 16


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32)]


class SiftParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("n_octave_layers", C.c_int32), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("sigma", C.c_double)]


class PairOpts(C.Structure):
    _fields_ = [("match_mode", C.c_int32), ("ratio", C.c_double), ("ransac_prob", C.c_double),
                ("ransac_thresh", C.c_double), ("ransac_max_iters", C.c_int32), ("ransac_seed", C.c_uint64),
                ("pose_dist_thresh", C.c_double), ("want_points", C.c_int32)]


PAIR_RESULT_DTYPE = np.dtype([("n_kp1", "<i4"), ("n_kp2", "<i4"), ("n_match", "<i4"), ("n_inl", "<i4"),
                              ("n_good", "<i4"), ("status", "<i4"), ("ransac_iters", "<i4"), ("reserved", "<i4"),
                              ("R", "<f8", (9,)), ("t", "<f8", (3,)), ("E", "<f8", (9,))], align=True)

_P = C.c_void_p
_SIGS = {
    "vo_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "vo_destroy": (None, [_P]),
    "vo_last_error": (C.c_char_p, [_P]),
    "vo_version": (C.c_int, []),
    "vo_orb_detect_and_compute": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "vo_match_hamming": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "vo_match_l2": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "vo_knn2_ratio_hamming": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_double, _P, _P, _P, _P]),
    "vo_find_essential_ransac": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_double, C.c_double, C.c_int, C.c_uint64, _P, _P, _P, _P]),
    "vo_recover_pose": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_double, _P, _P, _P, _P]),
    "vo_triangulate": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P]),
    "vo_packed_pyramid_bytes": (C.c_int64, [C.c_int, C.c_int, _P]),
    "vo_stage_pyramid": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "vo_stage_fast_scores": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "vo_stage_blur": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "vo_stage_five_point": (C.c_int, [_P, _P, _P, _P, _P]),
    "vo_batch_configure": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int]),
    "vo_frames_upload": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int]),
    "vo_frames_upload_async": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int]),
    "vo_frames_upload_color": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]),
    "vo_frames_detect": (C.c_int, [_P, C.c_int, C.c_int]),
    "vo_frames_detect_async": (C.c_int, [_P, C.c_int, C.c_int]),
    "vo_batch_kp_capacity": (C.c_int, [_P]),
    "vo_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "vo_host_free": (None, [_P]),
    "vo_frame_features": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "vo_pairs_run": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, C.c_int32]),
    "vo_pairs_run_async": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, C.c_int32]),
    "vo_sync": (C.c_int, [_P]),
    "vo_set_matcher_kernel": (C.c_int, [_P, C.c_int]),
    "vo_set_poly_solver": (C.c_int, [_P, C.c_int]),
    "vo_set_pnp_refine": (C.c_int, [_P, C.c_int]),
    "vo_set_keypoint_order": (C.c_int, [_P, C.c_int]),
    "vo_stage_retain_best": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "vo_detect_after": (C.c_int, [_P, _P]),
    "vo_knn2_hamming": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P]),
    "vo_knn2_l2": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "vo_knn2_ratio_l2": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P]),
    "vo_tracks_pnp_batch": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_double, _P, _P, _P, _P, _P]),
    "vo_comm_unique_id": (C.c_int, [_P]),
    "vo_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "vo_comm_destroy": (C.c_int, [_P]),
    "vo_comm_share": (C.c_int, [_P, _P]),
    "vo_comm_info": (C.c_int, [_P, _P, _P]),
    "vo_pairs_gather": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "vo_comm_allgather_f64": (C.c_int, [_P, _P, C.c_int, _P]),
    "vo_pair_matches": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "vo_reprojection_filter": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.c_double, _P, _P]),
    "vo_solve_pnp_ransac": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.c_double, C.c_double, C.c_uint64, _P, _P, _P, _P]),
    "vo_solve_pnp_ransac_batch": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_double, C.c_double, C.c_uint64, _P, _P, _P, _P, _P]),
    "vo_rodrigues": (C.c_int, [_P, _P, C.c_int, _P]),
    "vo_resize_linear": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "vo_resize_area": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "vo_frames_ingest": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _P]),
    "vo_sift_detect_and_compute": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "vo_batch_configure_sift": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "vo_frame_features_sift": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "vo_jpeg_info": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, _P]),
    "vo_jpeg_decode": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int, C.c_int, _P, _P]),
    "vo_jpeg_decode_batch": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.c_int]),
    "vo_frames_ingest_jpeg": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "vo_feature_tracks": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "vo_profile_enable": (C.c_int, [_P, C.c_int]),
    "vo_profile_reset": (C.c_int, [_P]),
    "vo_profile_read": (C.c_int, [_P, _P, _P]),
    "vo_stage_name": (C.c_char_p, [C.c_int]),
    "vo_stage_bytes": (C.c_double, [_P, C.c_int, C.c_int]),
}

_lib = None


def load():
    """Load libvo_hip.so and declare every entry point. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C visual_odometry_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvo_hip error {code}: {msg}")
        self.code = code


class Context:
    """One vo_ctx: owns device buffers and a HIP stream on `device`. Not thread safe."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _P()
        rc = self.lib.vo_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise VoError(rc, f"vo_create(device={device}) failed — is a gfx950 GPU visible to this process?")
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_matcher_kernel(self, kind):
        """'mfma_fp4' (default: block-scaled FP4 matrix cores), 'mfma' (int8 matrix cores) or 'popcount': which kernel computes
        the Hamming nearest neighbours (same results).  Choose before detecting."""
        self.check(self.lib.vo_set_matcher_kernel(self.handle, {"mfma": 0, "popcount": 1, "mfma_fp4": 2}[kind]))

    # ---- multi-GPU: the trajectory gather over RCCL (one communicator per process, shared by its contexts)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * VO_COMM_ID_BYTES)()
        if self.lib.vo_comm_unique_id(buf) != 0:
            raise VoError(VO_ERR_HIP, "ncclGetUniqueId failed (is librccl.so.1 loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * VO_COMM_ID_BYTES).from_buffer_copy(unique_id)
        # RCCL prints a version banner on the process's STDOUT when the first communicator is created; a launcher's stdout is a
        # data channel (bench.py prints one JSON line), so the banner is sent to stderr: fd 1 points at fd 2 for the call
        import sys
        sys.stdout.flush()
        keep = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = self.lib.vo_comm_init(self.handle, buf, int(rank), int(world))
        finally:
            os.dup2(keep, 1)
            os.close(keep)
        self.check(rc)

    def comm_destroy(self):
        self.check(self.lib.vo_comm_destroy(self.handle))

    def comm_share(self, owner):
        """Join the communicator `owner` (another Context of this process and GPU) created: one communicator per process."""
        self.check(self.lib.vo_comm_share(self.handle, owner.handle))

    def comm_info(self):
        """(ranks in the communicator as ncclCommCount reports them, this process's rank); (1, 0) without a communicator."""
        n, r = C.c_int32(0), C.c_int32(0)
        self.check(self.lib.vo_comm_info(self.handle, C.addressof(n), C.addressof(r)))
        return n.value, r.value

    def allgather(self, values, world):
        """Synchronous all-gather of a few float64 per rank over the context's communicator -> [world, n]."""
        v = np.ascontiguousarray(values, np.float64).ravel()
        out = np.empty((int(world), len(v)), np.float64)
        self.check(self.lib.vo_comm_allgather_f64(self.handle, v.ctypes.data, len(v), out.ctypes.data))
        return out

    def set_keypoint_order(self, kind):
        """'cv2' (default): the order cv2.ORB returns the keypoints in; 'canonical': (octave, y, x) order, the same set;
        (KeyPointsFilter::retainBest's libstdc++ permutation), so keypoint and match indices equal cv2's."""
        self.check(self.lib.vo_set_keypoint_order(self.handle, {"canonical": 0, "cv2": 1}[kind]))

    def retain_best(self, response, n_points):
        """cv::KeyPointsFilter::retainBest on a response list: the kept original indices in cv2's order."""
        r = np.ascontiguousarray(response, np.float32)
        out = np.zeros(max(len(r), 1), np.int32)
        n = C.c_int32(0)
        self.check(self.lib.vo_stage_retain_best(self.handle, r.ctypes.data, len(r), int(n_points), out.ctypes.data, C.addressof(n)))
        return out[:n.value].copy()

    def set_poly_solver(self, kind):
        """'fast' (default): Durand-Kerner sweeps stop at the rounding-noise floor; 'opencv300': cv::solvePoly's fixed
        300 sweeps (the faithful, 10x slower form of the five-point solver's root finder)."""
        self.check(self.lib.vo_set_poly_solver(self.handle, {"fast": 0, "opencv300": 1}[kind]))

    def set_pnp_refine(self, kind):
        """'cv2' (default): solvePnPRansac's final pose as OpenCV computes it (DLT / homography start + CvLevMarq, <= 20
        iterations); 'fast': the same cost minimised from the best RANSAC model."""
        self.check(self.lib.vo_set_pnp_refine(self.handle, {"fast": 0, "cv2": 1}[kind]))

    def last_error(self) -> str:
        msg = self.lib.vo_last_error(self.handle)
        return msg.decode() if msg else ""

    def check(self, rc):
        if rc < 0:
            msg = self.lib.vo_last_error(self.handle)
            raise VoError(rc, msg.decode() if msg else "")
        return rc


_default = {}


def default_context(device: int = 0) -> Context:
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


def ptr(a):
    return None if a is None else a.ctypes.data


class PinnedArray:
    """numpy view over page-locked host memory from vo_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype):
        lib = load()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = _P()
        if lib.vo_host_alloc(max(n, 1), C.byref(p)) != 0 or not p:
            raise MemoryError(f"vo_host_alloc({n}) failed")
        self._lib, self._ptr = lib, p
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            if self._ptr:
                self.array = None
                self._lib.vo_host_free(self._ptr)
                self._ptr = None
        except Exception:
            pass
