"""Seeded synthetic drone sequences (the reference ships no images: its .gitignore:1-4 excludes
input/*, the DJI video and frames).  Build-owned; used by tests/ and bench.py.

Scene: a 4096x4096 u8 ground texture (random rectangles and discs plus noise) on the plane
z = 0, with box-shaped relief of up to 10 % of the flight height so the scene is not planar.
Camera: nadir pin-hole at height 30, intrinsics = the reference's (visual_slam.py:30-36:
f = 2676.105*0.3 at 1152 px width) rescaled to the frame width; it advances +x by `step`
per frame with a small yaw and seeded pitch jitter.  Frames are rendered by a plane sweep
(highest relief level first, inverse mapping, bilinear sampling) in numpy.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

TEX_SIZE = 4096
PPU = 32.0            # texture pixels per world unit
HEIGHT = 30.0
RELIEF = (3.0, 2.0, 1.0, 0.0)
_SEED = 20260104

_cache = {}


def camera_matrix(w: int, h: int) -> np.ndarray:
    f = 0.3 * 2676.105 * (w / 1152.0)
    return np.array([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]], dtype=np.float64)


def reference_camera_matrix() -> np.ndarray:
    """visual_slam.py:28-37 set_camera_matrix (known answer: test.g2o:1 = 802.832 565.427 240.124)."""
    k = np.array([[2676.1051390718389, 0.0, 3840 / 2 - 35.243952918157035],
                  [0.0, 2676.1051390718389, 2160 / 2 - 279.58562078697361],
                  [0.0, 0.0, 1.0]])
    k = k * 0.3
    k[2, 2] = 1
    return k


def ground_texture(seed: int = _SEED):
    """Returns (texture u8 [S,S], heightmap f32 [S,S])."""
    key = ("tex", seed)
    if key in _cache:
        return _cache[key]
    rng = np.random.default_rng(seed)
    s = TEX_SIZE
    tex = np.full((s, s), 110.0, dtype=np.float32)
    n = 9000
    cx = rng.integers(0, s, n); cy = rng.integers(0, s, n)
    sz = rng.integers(6, 61, n); sz2 = rng.integers(6, 61, n)
    grey = rng.integers(0, 256, n); kind = rng.integers(0, 2, n)
    for i in range(n):
        x0, y0 = max(cx[i] - sz[i] // 2, 0), max(cy[i] - sz2[i] // 2, 0)
        x1, y1 = min(cx[i] + sz[i] // 2 + 1, s), min(cy[i] + sz2[i] // 2 + 1, s)
        if kind[i] == 0:
            tex[y0:y1, x0:x1] = grey[i]
        else:
            yy, xx = np.ogrid[y0:y1, x0:x1]
            r = sz[i] / 2.0
            m = (xx - cx[i]) ** 2 + (yy - cy[i]) ** 2 <= r * r
            tex[y0:y1, x0:x1][m] = grey[i]
    tex += rng.normal(0.0, 6.0, (s, s)).astype(np.float32)
    tex = np.clip(tex, 0, 255).astype(np.uint8)
    hmap = np.zeros((s, s), dtype=np.float32)
    nb = 260
    bx = rng.integers(0, s, nb); by = rng.integers(0, s, nb)
    bw = rng.integers(60, 260, nb); bh = rng.integers(60, 260, nb)
    lev = rng.integers(1, len(RELIEF), nb)
    for i in range(nb):
        x0, y0 = max(bx[i] - bw[i] // 2, 0), max(by[i] - bh[i] // 2, 0)
        x1, y1 = min(bx[i] + bw[i] // 2, s), min(by[i] + bh[i] // 2, s)
        hmap[y0:y1, x0:x1] = np.maximum(hmap[y0:y1, x0:x1], RELIEF[len(RELIEF) - 1 - lev[i]])
    _cache[key] = (tex, hmap)
    return tex, hmap


def poses(n_frames: int, step: float = 1.0, yaw_deg: float = 0.5, seed: int = _SEED):
    """World-to-camera rotations R (n,3,3) and camera centres C (n,3); x_cam = R (X - C)."""
    rng = np.random.default_rng(seed + 1)
    pitch = rng.uniform(-0.2, 0.2, n_frames) * np.pi / 180.0
    r0 = np.diag([1.0, -1.0, -1.0])
    rs, cs = [], []
    x_start = -0.5 * step * (n_frames - 1)
    for k in range(n_frames):
        yaw = np.deg2rad(yaw_deg * k)
        cz, sz = np.cos(yaw), np.sin(yaw)
        rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
        cp, sp = np.cos(pitch[k]), np.sin(pitch[k])
        rx = np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
        rs.append(rx @ r0 @ rz.T)
        cs.append(np.array([x_start + step * k, 0.0, HEIGHT]))
    return np.stack(rs), np.stack(cs)


def loop_poses(n_frames: int, radius: float = 36.0, seed: int = _SEED):
    """A closed circular flight of n_frames views (frame n would be frame 0 again): heading along the tangent, so
    consecutive frames differ by an arc of 2 pi radius / n and a yaw of 360 / n degrees, with the same seeded pitch
    jitter as the straight path.  The whole footprint stays inside the texture, however long the sequence."""
    rng = np.random.default_rng(seed + 2)
    pitch = rng.uniform(-0.2, 0.2, n_frames) * np.pi / 180.0
    r0 = np.diag([1.0, -1.0, -1.0])
    rs, cs = [], []
    for k in range(n_frames):
        th = 2.0 * np.pi * k / n_frames
        yaw = th + np.pi / 2.0                                   # image x axis along the direction of flight
        cz, sz = np.cos(yaw), np.sin(yaw)
        rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
        cp, sp = np.cos(pitch[k]), np.sin(pitch[k])
        rx = np.array([[1.0, 0, 0], [0, cp, -sp], [0, sp, cp]])
        rs.append(rx @ r0 @ rz.T)
        cs.append(np.array([radius * np.cos(th), radius * np.sin(th), HEIGHT]))
    return np.stack(rs), np.stack(cs)


def relative_pose(r1, c1, r2, c2):
    """R, t_hat with x2 ~ R x1 + t (the convention cv2.recoverPose returns)."""
    r = r2 @ r1.T
    t = r2 @ (c1 - c2)
    n = np.linalg.norm(t)
    return r, (t / n if n > 0 else t)


def _bilinear(tex, u, v):
    s = tex.shape[0]
    u = np.clip(u, 0, s - 1.001); v = np.clip(v, 0, s - 1.001)
    iu = u.astype(np.int32); iv = v.astype(np.int32)
    fu = u - iu; fv = v - iv
    t = tex
    a = t[iv, iu].astype(np.float32); b = t[iv, iu + 1].astype(np.float32)
    c = t[iv + 1, iu].astype(np.float32); d = t[iv + 1, iu + 1].astype(np.float32)
    return (a * (1 - fu) + b * fu) * (1 - fv) + (c * (1 - fu) + d * fu) * fv


def render_frame(w, h, k, r, c, seed=_SEED, noise_seed=0):
    tex, hmap = ground_texture(seed)
    kinv = np.linalg.inv(k)
    px, py = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    m = (r.T @ kinv).astype(np.float32)
    dx = m[0, 0] * px + m[0, 1] * py + m[0, 2]
    dy = m[1, 0] * px + m[1, 1] * py + m[1, 2]
    dz = m[2, 0] * px + m[2, 1] * py + m[2, 2]
    out = np.zeros((h, w), dtype=np.float32)
    done = np.zeros((h, w), dtype=bool)
    half = TEX_SIZE / 2.0
    for lev in RELIEF:
        s = (lev - c[2]) / dz
        u = (c[0] + s * dx) * PPU + half
        v = (c[1] + s * dy) * PPU + half
        ui = np.clip(u, 0, TEX_SIZE - 1).astype(np.int32)
        vi = np.clip(v, 0, TEX_SIZE - 1).astype(np.int32)
        hit = (hmap[vi, ui] >= lev - 1e-3) & ~done
        if hit.any():
            out[hit] = _bilinear(tex, u[hit], v[hit])
            done |= hit
    rng = np.random.default_rng(seed * 7919 + noise_seed + 17)
    out += rng.normal(0.0, 1.5, out.shape).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def _render_job(args):
    w, h, k, r, c, seed, i = args
    return render_frame(w, h, k, r, c, seed, i)


def sequence(n_frames: int, w: int, h: int, step: float = 1.0, yaw_deg: float = 0.5,
             seed: int = _SEED, cache_dir: str | None = None, trajectory: str = "line", workers: int | None = None):
    """Returns dict(frames u8 [n,h,w], K, R [n,3,3], C [n,3]).  trajectory = "line": the straight flight (step / yaw per
    frame); "loop": the closed circular flight of loop_poses (n distinct views, frame n == frame 0)."""
    k = camera_matrix(w, h)
    rs, cs = loop_poses(n_frames, seed=seed) if trajectory == "loop" else poses(n_frames, step, yaw_deg, seed)
    path = cache_path(n_frames, w, h, cache_dir, trajectory, step, yaw_deg, seed) if cache_dir else None
    if path and os.path.exists(path):
        try:
            frames = np.load(path)
            if frames.shape == (n_frames, h, w):
                return dict(frames=frames, K=k, R=rs, C=cs)
        except (OSError, ValueError):
            pass
    ground_texture(seed)                         # built once, inherited by forked render workers
    if workers is None:
        workers = min(os.cpu_count() or 1, 16) if n_frames >= 32 else 1
    jobs = [(w, h, k, rs[i], cs[i], seed, i) for i in range(n_frames)]
    if workers > 1:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(workers)
        try:
            frames = np.stack(pool.map(_render_job, jobs, chunksize=2))
        finally:
            pool.close()                         # let the workers leave by themselves: Pool.__exit__ is terminate() = SIGTERM,
            pool.join()                          # which a tool's signal handler inherited by a worker reports as an abort
    else:
        frames = np.stack([_render_job(j) for j in jobs])
    if path:
        try:                                   # atomic publish: several ranks may render the same sequence at once
            tmp = f"{path}.{os.getpid()}.tmp.npy"
            np.save(tmp, frames)
            os.replace(tmp, path)
        except OSError:
            pass
    return dict(frames=frames, K=k, R=rs, C=cs)


def cache_path(n_frames: int, w: int, h: int, cache_dir: str, trajectory: str = "line", step: float = 1.0, yaw_deg: float = 0.5,
               seed: int = _SEED):
    if trajectory == "loop":
        tag = hashlib.sha1(f"loop-{n_frames}-{w}-{h}-{seed}-v1".encode()).hexdigest()[:16]
    else:
        tag = hashlib.sha1(f"{n_frames}-{w}-{h}-{step}-{yaw_deg}-{seed}-v1".encode()).hexdigest()[:16]
    return os.path.join(cache_dir, f"vo_synth_{tag}.npy")


def _cache_complete(path, shape):
    try:
        return np.load(path, mmap_mode="r").shape == tuple(shape)
    except (OSError, ValueError):
        return False


# what a profiler or sanitizer injects into a process it wraps: the render child is CPU-only numpy and must not carry it
_TOOL_ENV_PREFIXES = ("LD_PRELOAD", "ROCP", "ROCPROF", "ROCTX", "HSA_TOOLS", "ROCTRACER", "AMD_LOG", "OMPI_", "ASAN_OPTIONS")


def prerender(n_frames: int, w: int, h: int, cache_dir: str = "/tmp", trajectory: str = "line"):
    """Fill the cache for sequence(...) in a CHILD process (python -m visual_odometry_amd.synth): the parallel renderer
    forks workers, which a process that holds (or will hold) a GPU context and a process group should not do itself.
    Call it before anything initialises the GPU; afterwards sequence(..., workers=1) only reads the cache.
    Nothing is started when the cache is complete, and the child's environment is stripped of profiler / tool preloads
    (under `rocprofv3 -- python3 bench.py` the child and its forked workers would otherwise carry the profiler)."""
    import subprocess
    import sys
    if _cache_complete(cache_path(n_frames, w, h, cache_dir, trajectory), (n_frames, h, w)):
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith(_TOOL_ENV_PREFIXES)}
    subprocess.check_call([sys.executable, "-m", "visual_odometry_amd.synth", str(n_frames), str(w), str(h), cache_dir, trajectory],
                          cwd=root, stdout=subprocess.DEVNULL, env=env)


if __name__ == "__main__":
    import sys
    _n, _w, _h, _dir, _traj = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    sequence(_n, _w, _h, cache_dir=_dir, trajectory=_traj)
