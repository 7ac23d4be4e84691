"""Batched, device-resident front end: frames live in HBM, detection + matching + E-RANSAC + pose (+ DLT) run
batch-major through the C ABI with no host round trip between stages.  This is the throughput path
bench.py measures; the per-pair order is that of src/visual_slam.py:294-298.  detector="orb" is the north-star
instantiation (ORB + Hamming, src/image_and_keypoints.py:8-9); detector="sift" is the configuration the reference runs
live (cv2.SIFT_create() + BFMatcher(NORM_L2, crossCheck=True), src/visual_slam.py:17,19)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .detector import make_params
from .geometry import OPENCV_RNG_SEED

MATCH_CROSSCHECK, MATCH_RATIO, MATCH_CROSSCHECK_LEGACY = 0, 1, 2


class FrontEnd:
    def __init__(self, height, width, max_frames, max_pairs, nfeatures=500, nlevels=8, device=0, ctx=None,
                 keypoint_order="cv2", detector="orb", kp_cap=0, **det_kw):
        self.ctx = ctx or _lib.Context(device)
        self.h, self.w = int(height), int(width)
        self.max_frames, self.max_pairs = int(max_frames), int(max_pairs)
        self.detector = detector
        c = self.ctx
        if detector == "sift":
            # cv2.SIFT_create(nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04, edgeThreshold=10, sigma=1.6); kp_cap: keypoints
            # kept per frame (0 = a default from the frame size; more are truncated and flagged)
            self.params = _lib.SiftParams(0, int(det_kw.pop("nOctaveLayers", 3)), float(det_kw.pop("contrastThreshold", 0.04)),
                                          float(det_kw.pop("edgeThreshold", 10.0)), float(det_kw.pop("sigma", 1.6)))
            if det_kw:
                raise TypeError(f"unknown SIFT arguments {sorted(det_kw)}")
            rc = c.lib.vo_batch_configure_sift(c.handle, self.h, self.w, C.addressof(self.params), self.max_frames, self.max_pairs, int(kp_cap))
            if rc == _lib.VO_ERR_UNSUPPORTED:
                raise NotImplementedError(c.last_error())
            c.check(rc)
        elif detector == "orb":
            self.ctx.set_keypoint_order(keypoint_order)       # 'cv2': keypoint / match indices as cv2.ORB + BFMatcher number them
            self.params = make_params(nfeatures=nfeatures, nlevels=nlevels, **det_kw)
            c.check(c.lib.vo_batch_configure(c.handle, self.h, self.w, C.addressof(self.params), self.max_frames,
                                             self.max_pairs))
        else:
            raise ValueError("detector must be 'orb' or 'sift'")
        self.kp_cap = int(c.lib.vo_batch_kp_capacity(c.handle))
        self._res = _lib.PinnedArray((self.max_pairs,), _lib.PAIR_RESULT_DTYPE)      # page-locked result buffers,
        self._X = None                                                                # reused by every run_pairs call

    def pinned_frames(self, count):
        """Page-locked [count, H, W] uint8 staging array for upload(..., wait=False) (kept alive by the caller)."""
        return _lib.PinnedArray((int(count), self.h, self.w), np.uint8)

    def upload(self, frames, first_slot=0, wait=True):
        """frames: [F, H, W] gray or [F, H, W, 3|4] BGR(A) uint8 (BGR is converted on the device, as ORB does).
        wait=False (gray only): enqueue the copy; `frames` should come from pinned_frames() and must not change
        before the next wait()."""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        if f.ndim == 2 or (f.ndim == 3 and f.shape[-1] in (3, 4) and f.shape[:2] == (self.h, self.w)):
            f = f[None]
        if f.ndim not in (3, 4) or f.shape[1:3] != (self.h, self.w):
            raise ValueError(f"frames must be [F, {self.h}, {self.w}] or [F, {self.h}, {self.w}, 3|4] uint8")
        c = self.ctx
        if f.ndim == 3 and not wait:
            self._keep_frames = f
            c.check(c.lib.vo_frames_upload_async(c.handle, f.ctypes.data, f.shape[0], f.strides[1], f.strides[0], int(first_slot)))
        elif f.ndim == 3:
            c.check(c.lib.vo_frames_upload(c.handle, f.ctypes.data, f.shape[0], f.strides[1], f.strides[0], int(first_slot)))
        else:
            c.check(c.lib.vo_frames_upload_color(c.handle, f.ctypes.data, f.shape[0], f.shape[3], f.strides[1], f.strides[0],
                                                 int(first_slot)))

    def ingest(self, frames, first_slot=0, want_resized=False):
        """Full-resolution frames [F, H0, W0] or [F, H0, W0, 3|4] uint8 (as cv2.imread returns them): resized on the
        device to this front end's (w, h) with cv2.resize's default INTER_LINEAR (visual_slam.py:346-352), converted
        to gray as ORB does, stored as level 0 of the slots.  Returns the resized frames if want_resized."""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        if f.ndim == 2 or (f.ndim == 3 and f.shape[-1] in (3, 4) and f.shape[1] > 4):
            f = f[None]
        if f.ndim not in (3, 4):
            raise ValueError("frames must be [F, H, W] or [F, H, W, 3|4] uint8")
        cn = 1 if f.ndim == 3 else f.shape[3]
        out = None
        if want_resized:
            out = np.empty((f.shape[0], self.h, self.w) if f.ndim == 3 else (f.shape[0], self.h, self.w, cn), np.uint8)
        c = self.ctx
        c.check(c.lib.vo_frames_ingest(c.handle, f.ctypes.data, f.shape[0], f.shape[1], f.shape[2], cn, f.strides[1],
                                       f.strides[0], int(first_slot), _lib.ptr(out)))
        return out

    def ingest_jpeg(self, buffers, first_slot=0, want_resized=False):
        """buffers: a sequence of bytes-like objects or an ingest.PackedFiles (files packed in page-locked memory).
        The reference's whole ingest (visual_slam.py:346-352) for JPEG files of one size: cv2.imread -> cv2.resize to this
        front end's (w, h) -> gray into level 0 of the slots, all on the device (only the compressed bytes cross PCIe).
        Returns the resized B G R frames if want_resized (the reference keeps them as Frame.image)."""
        from .ingest import _packed
        blob, offs, _keep = _packed(buffers)                 # a PackedFiles (page-locked) goes over PCIe by DMA as it is
        n = len(offs) - 1
        out = np.empty((n, self.h, self.w, 3), np.uint8) if want_resized else None
        c = self.ctx
        rc = c.lib.vo_frames_ingest_jpeg(c.handle, blob.ctypes.data, offs.ctypes.data, n, int(first_slot), _lib.ptr(out))
        if rc == _lib.VO_ERR_UNSUPPORTED:
            raise NotImplementedError(c.last_error())
        c.check(rc)
        return out

    def detect(self, first_slot, count, wait=True, after=None):
        """ORB detect + describe of `count` resident slots. wait=False only enqueues the work on the ctx stream;
        the next run_pairs (same stream) is ordered after it.  after=<another FrontEnd on this GPU>: start only when
        that one's latest asynchronous detection has finished (keeps two alternating contexts out of phase)."""
        c = self.ctx
        if after is not None and after is not self:
            c.check(c.lib.vo_detect_after(c.handle, after.ctx.handle))
        fn = c.lib.vo_frames_detect if wait else c.lib.vo_frames_detect_async
        self._warn_capacity(c.check(fn(c.handle, int(first_slot), int(count))))

    def features(self, slot):
        """Keypoints and descriptors of a detected slot.  ORB: desc [n, 32] uint8; SIFT: desc [n, 128] float32 (the integer bin
        values 0..255 cv2 returns as floats)."""
        cap = self.kp_cap
        sift = self.detector == "sift"
        xy = np.empty((cap, 2), np.float32); size = np.empty(cap, np.float32); ang = np.empty(cap, np.float32)
        resp = np.empty(cap, np.float32); octv = np.empty(cap, np.int32); desc = np.empty((cap, 128 if sift else 32), np.uint8)
        n = C.c_int32(0)
        c = self.ctx
        fn = c.lib.vo_frame_features_sift if sift else c.lib.vo_frame_features
        rc = c.check(fn(c.handle, int(slot), xy.ctypes.data, size.ctypes.data, ang.ctypes.data,
                        resp.ctypes.data, octv.ctypes.data, desc.ctypes.data, cap, C.addressof(n)))
        k = n.value
        d = desc[:k].astype(np.float32) if sift else desc[:k].copy()
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=ang[:k].copy(), response=resp[:k].copy(),
                    octave=octv[:k].copy(), desc=d, truncated=(rc == _lib.VO_WARN_CAPACITY))

    def make_opts(self, match_mode=MATCH_CROSSCHECK, ratio=0.75, prob=0.99, thresh=1.0, max_iters=1000,
                  seed=OPENCV_RNG_SEED, dist_thresh=50.0, want_points=False):
        return _lib.PairOpts(int(match_mode), float(ratio), float(prob), float(thresh), int(max_iters), int(seed),
                             float(dist_thresh), int(bool(want_points)))

    def run_pairs(self, pair_slots, K, opts=None, want_points=False, wait=True):
        """pair_slots: [B, 2] int32 of detected slots. Returns (results structured array [B], X or None) — views of
        reused page-locked buffers (copy them if they must outlive the next call).  wait=False only enqueues the
        work; the returned views are valid after self.wait()."""
        ps = np.ascontiguousarray(pair_slots, dtype=np.int32).reshape(-1, 2)
        B = len(ps)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        opts = opts or self.make_opts(want_points=want_points)
        if B > self.max_pairs:
            raise ValueError(f"{B} pairs > max_pairs={self.max_pairs}")
        res = self._res.array[:B]
        X = None
        if opts.want_points:
            if self._X is None:
                self._X = _lib.PinnedArray((self.max_pairs, 4, self.kp_cap), np.float64)
            X = self._X.array[:B]
        c = self.ctx
        fn = c.lib.vo_pairs_run if wait else c.lib.vo_pairs_run_async
        self._keep = (ps, K, opts)                  # keep the argument buffers alive until the work has been consumed
        self._warn_capacity(c.check(fn(c.handle, ps.ctypes.data, B, K.ctypes.data, C.addressof(opts), res.ctypes.data, _lib.ptr(X), self.kp_cap)))
        return res, X

    def _warn_capacity(self, rc):
        if rc == _lib.VO_WARN_CAPACITY:
            import warnings
            warnings.warn("a frame's keypoint list hit its capacity and was cut (SIFT: in x order, the right edge of the image first): "
                          "raise kp_cap; features(slot)['truncated'] names the slots", RuntimeWarning, stacklevel=3)

    def wait(self):
        c = self.ctx
        c.check(c.lib.vo_sync(c.handle))

    def localize_chain(self, n_pairs, K, iterations=100, reproj_err=8.0, confidence=0.99, seed=OPENCV_RNG_SEED, max_point_norm=50.0):
        """The step after the pair path on resident data (vo_tracks_pnp_batch; src/visual_slam.py:183-266 without the bundle
        adjustment): the `n_pairs` pairs of the latest run_pairs(..., want_points=True) — a chain (f0, f1), (f1, f2), ... — are
        walked in order: feature tracks -> map / image coordinates -> solvePnPRansac -> camera -> new map points.
        Returns dict(poses [n_pairs + 1, 3, 4] world -> camera, n_corr, n_inl, status, n_map, each [n_pairs])."""
        B = int(n_pairs)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        poses = np.zeros((B + 1, 12)); nc = np.zeros(B, np.int32); ni = np.zeros(B, np.int32); st = np.zeros(B, np.int32); nm = np.zeros(B, np.int32)
        c = self.ctx
        c.check(c.lib.vo_tracks_pnp_batch(c.handle, B, K.ctypes.data, int(iterations), float(reproj_err), float(confidence), int(seed),
                                          float(max_point_norm), poses.ctypes.data, nc.ctypes.data, ni.ctypes.data, st.ctypes.data, nm.ctypes.data))
        return dict(poses=poses.reshape(B + 1, 3, 4), n_corr=nc, n_inl=ni, status=st, n_map=nm)

    def gather_records(self, B, world=1, wait=True):
        """All-gather the [R|t] + counts records (16 float64 per pair) of the first B pairs of the latest run_pairs
        over the context's RCCL communicator (ctx.comm_init; without one: the local records).  Returns a
        [world, B, 16] view of a reused page-locked buffer, valid at once (wait) or after self.wait()."""
        if getattr(self, "_gath", None) is None or self._gath.array.shape[0] < world * self.max_pairs:
            self._gath = _lib.PinnedArray((world * self.max_pairs, _lib.VO_RECORD_DOUBLES), np.float64)
        out = self._gath.array[:world * B]
        c = self.ctx
        c.check(c.lib.vo_pairs_gather(c.handle, int(B), out.ctypes.data, int(bool(wait))))
        return out.reshape(world, B, _lib.VO_RECORD_DOUBLES)

    def pair_matches(self, pair):
        cap = self.kp_cap
        qi = np.empty(cap, np.int32); ti = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
        m = np.empty(cap, np.uint8); n = C.c_int32(0)
        c = self.ctx
        c.check(c.lib.vo_pair_matches(c.handle, int(pair), qi.ctypes.data, ti.ctypes.data, d.ctypes.data,
                                      m.ctypes.data, cap, C.addressof(n)))
        k = n.value
        return qi[:k].copy(), ti[:k].copy(), d[:k].copy(), m[:k].copy()

    # ---- measurement ------------------------------------------------------------------------
    def profile(self, on=True):
        c = self.ctx
        c.check(c.lib.vo_profile_enable(c.handle, int(on)))
        c.check(c.lib.vo_profile_reset(c.handle))

    def profile_read(self):
        ms = np.zeros(_lib.VO_STAGE_COUNT, np.float32); n = np.zeros(_lib.VO_STAGE_COUNT, np.int32)
        c = self.ctx
        c.check(c.lib.vo_profile_read(c.handle, ms.ctypes.data, n.ctypes.data))
        names = [c.lib.vo_stage_name(i).decode() for i in range(_lib.VO_STAGE_COUNT)]
        return {names[i]: (float(ms[i]), int(n[i])) for i in range(_lib.VO_STAGE_COUNT) if n[i] > 0}

    def stage_bytes(self, stage_name, frames):
        c = self.ctx
        for i in range(_lib.VO_STAGE_COUNT):
            if c.lib.vo_stage_name(i).decode() == stage_name:
                return float(c.lib.vo_stage_bytes(c.handle, i, int(frames)))
        raise KeyError(stage_name)


def chain_poses(R, t):
    """Compose relative poses x_{k+1} ~ R_k x_k + t_k into camera-to-world 4x4 matrices (unit-norm t: the
    scale of every step is unobservable, as in the reference's monocular front end)."""
    T = np.eye(4)
    out = [T.copy()]
    for Rk, tk in zip(R, t):
        step = np.eye(4)
        step[:3, :3] = Rk
        step[:3, 3] = np.asarray(tk).ravel()
        T = T @ np.linalg.inv(step)
        out.append(T.copy())
    return np.stack(out)
