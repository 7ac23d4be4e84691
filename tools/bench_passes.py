"""Timed passes over the rows either side of the ORB pair path, shared by bench.py (its `config.sift`, `config.jpeg_pipeline`,
`config.pnp` entries: the widened rows under the driver's clock) and by the stand-alone scripts in tests/scripts/.  Product
code only: nothing here touches the CPU oracle.

  jpeg_files()          seeded views -> baseline JPEG files in memory (CPU, Pillow's encoder; before anything initialises the GPU)
  jpeg_pipeline_pass()  files in -> poses out: vo_frames_ingest_jpeg (decode + cv2.resize + gray on the device) -> detect ->
                        pairs, several contexts on host threads (src/visual_slam.py:346-352 -> :294-298)
  sift_pass()           the reference's live configuration (cv2.SIFT_create() + BFMatcher(NORM_L2, crossCheck), :17,19) batched
  pnp_pass()            cv2.solvePnPRansac problems per second (:231-235)
"""
from __future__ import annotations

import io
import threading
import time

import numpy as np


def jpeg_files(frames, quality=90, chroma="scene"):
    """frames [n, h, w] gray views -> n JPEG files (4:2:0).  chroma "flat": the gray view in all three channels (Cb = Cr = 128
    everywhere: the hard case for the parallel entropy decoder); "scene": low-frequency chroma derived from the view itself,
    as a colour camera's files have."""
    from PIL import Image
    files = []
    for g in frames:
        if chroma == "flat":
            rgb = np.stack([g, g, g], -1)
        else:
            gf = g.astype(np.float32)
            lp = gf
            for _ in range(4):
                lp = (np.roll(lp, 8, 0) + np.roll(lp, -8, 0) + np.roll(lp, 8, 1) + np.roll(lp, -8, 1) + 4 * lp) / 8
            cb = 128 + 0.35 * (lp - 128) + 20 * np.sin(lp / 17.0); cr = 128 - 0.25 * (lp - 128) + 20 * np.cos(lp / 23.0)
            rgb = np.stack([gf + 1.402 * (cr - 128), gf - 0.344136 * (cb - 128) - 0.714136 * (cr - 128), gf + 1.772 * (cb - 128)], -1).clip(0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", quality=quality, subsampling=2)
        files.append(b.getvalue())
    return files


def jpeg_pipeline_pass(files, width, height, K, scale=1.0, detector="orb", pairs=256, contexts=3, chunks=12, nfeatures=2000, device=0, kp_cap=0):
    """A chunk = pairs + 1 JPEG files (width x height) -> ingest_jpeg (decode, resize x scale, gray) -> detect -> `pairs`
    consecutive frame pairs; `contexts` host threads, each with its own context, run `chunks` chunks.  Only the compressed
    bytes cross PCIe."""
    from visual_odometry_amd import _lib, ingest
    from visual_odometry_amd.frontend import FrontEnd
    C = int(pairs)
    bufs = [files[k % len(files)] for k in range(C + 1)]
    dw, dh = int(round(width * scale)), int(round(height * scale))
    Ks = np.array(K, np.float64).copy(); Ks[:2] *= scale
    pair_idx = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
    fes, packed, opts = [], [], []
    for _ in range(contexts):
        if detector == "sift":
            fe = FrontEnd(dh, dw, max_frames=C + 1, max_pairs=C, detector="sift", kp_cap=kp_cap, ctx=_lib.Context(device))
        else:
            fe = FrontEnd(dh, dw, max_frames=C + 1, max_pairs=C, nfeatures=nfeatures, ctx=_lib.Context(device))
        pk = ingest.PackedFiles(bufs)
        fe.ingest_jpeg(pk); fe.detect(0, C + 1); fe.run_pairs(pair_idx, Ks, fe.make_opts())          # warm-up (allocations)
        fes.append(fe); packed.append(pk); opts.append(fe.make_opts())
    ok = [0] * contexts; inl = [0] * contexts; err = [None] * contexts

    def work(c):
        try:
            fe = fes[c]
            for _ in range(chunks):
                fe.ingest_jpeg(packed[c])
                fe.detect(0, C + 1, wait=False)
                rec = fe.run_pairs(pair_idx, Ks, opts[c])[0]
                ok[c] += int((rec["status"] == 0).sum()); inl[c] += int(rec["n_inl"].sum())
        except Exception as e:                                  # noqa: BLE001 — reported by the caller's thread
            err[c] = e

    th = [threading.Thread(target=work, args=(c,)) for c in range(contexts)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for e in err:
        if e is not None:
            raise e
    n = contexts * chunks * C
    out = {"what": f"{C + 1} JPEG files {width}x{height} ({sum(map(len, bufs)) / len(bufs) / 1024:.0f} KiB each, 4:2:0) per chunk -> decode -> "
                   f"resize x{scale:g} -> {dw}x{dh} -> " + ("SIFT + L2 cross-check" if detector == "sift" else f"ORB {nfeatures} + Hamming cross-check") +
                   f" -> {C} pairs (E-RANSAC, pose); {contexts} contexts x {chunks} chunks; only the compressed bytes cross PCIe",
           "detector": detector, "pairs_per_s": round(n / dt, 1), "frames_per_s": round(contexts * chunks * (C + 1) / dt, 1),
           "ms_per_chunk": round(dt / (contexts * chunks) * 1e3, 2), "pairs_ok_fraction": round(sum(ok) / n, 4),
           "mean_inliers": round(sum(inl) / max(sum(ok), 1), 1), "pcie_bytes_per_pair": int(sum(map(len, bufs)) / C)}
    for fe in fes:
        fe.ctx.close()
    return out


def sift_pass(frames, K, contexts=2, steps=4, device=0, kp_cap=0, hbm_peak_gbs=8000.0):
    """frames [C + 1, h, w] resident -> batched SIFT -> C pairs (L2 cross-check, E-RANSAC, pose, DLT), `steps` steps per
    context alternating; then one profiled step on a single context for the scale space's roofline entry."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd
    from visual_odometry_amd.pipeline import ChunkPipeline
    NF, h, w = frames.shape
    C = NF - 1
    pairs = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
    fes = [FrontEnd(h, w, max_frames=NF, max_pairs=C, device=device, detector="sift", kp_cap=kp_cap, ctx=_lib.Context(device)) for _ in range(contexts)]
    for f in fes:
        f.upload(frames)
    opts = fes[0].make_opts(want_points=True)
    pipe = ChunkPipeline(fes, K, opts)
    for _ in range(contexts):
        pipe.submit(pairs, NF)
    pipe.drain()
    n = steps * contexts
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit(pairs, NF)
    last = pipe.drain()[-1]
    dt = time.perf_counter() - t0
    res = last.results[:C]
    fe = fes[0]
    fe.profile(True)
    fe.detect(0, NF, wait=False)
    fe.run_pairs(pairs, K, opts)
    prof = fe.profile_read()
    fe.profile(False)
    out = {"what": f"cv2.SIFT_create() defaults + BFMatcher(NORM_L2, crossCheck) on {NF} resident {w}x{h} frames -> {C} pairs per step "
                   f"(the reference's live configuration, src/visual_slam.py:17,19, at its working size), {contexts} contexts",
           "pairs_per_s": round(C * n / dt, 1), "ms_per_step": round(1e3 * dt / n, 3), "pairs_ok_last_step": int((res["status"] == 0).sum()),
           "mean_keypoints": round(float(res["n_kp1"].mean()), 1), "mean_inliers": round(float(res["n_inl"].mean()), 1)}
    rl = {}
    for k in ("sift_scale_space", "sift_extrema"):
        if k in prof and prof[k][0] > 0:
            b = fe.stage_bytes(k, NF)
            ach = b / (prof[k][0] * 1e-3) / 1e9
            rl[k] = {"bound": "hbm", "achieved": round(ach, 1), "peak": hbm_peak_gbs, "unit": "GB/s", "frac": round(ach / hbm_peak_gbs, 4),
                     "algorithmic_bytes_per_step": b, "ms_per_step": round(prof[k][0], 3), "launches_per_step": prof[k][1]}
    out["roofline"] = rl
    out["stages_ms_per_step"] = {k: round(v[0], 3) for k, v in prof.items()}
    for f in fes:
        f.ctx.close()
    return out


def chain_pass(frames, K, nfeatures=2000, device=0, repeats=3):
    """frames [n, h, w] -> ORB pairs (k, k + 1) with triangulation -> vo_tracks_pnp_batch: the reference's per-frame localisation
    (tracks -> map / image coordinates -> solvePnPRansac -> camera -> new map points, src/visual_slam.py:183-266 without BA) walked
    over the resident results.  The walk is sequential by nature (frame k + 1 is localised against the map frame k extended): five
    small launches per pair on one stream; sequences, not pairs, are what runs in parallel (contexts / GPUs)."""
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd
    n, h, w = frames.shape
    pairs = np.stack([np.arange(n - 1), np.arange(n - 1) + 1], 1).astype(np.int32)
    fe = FrontEnd(h, w, max_frames=n, max_pairs=n - 1, nfeatures=nfeatures, ctx=_lib.Context(device))
    fe.upload(frames); fe.detect(0, n)
    fe.run_pairs(pairs, K, want_points=True)
    out = fe.localize_chain(n - 1, K)                                   # warm-up (allocations)
    t0 = time.perf_counter()
    for _ in range(repeats):
        out = fe.localize_chain(n - 1, K)
    dt = (time.perf_counter() - t0) / repeats
    ok = int((out["status"] == 0).sum())
    res = {"what": f"vo_tracks_pnp_batch over {n - 1} consecutive {w}x{h} ORB pairs resident in HBM (src/visual_slam.py:183-266 without the bundle "
                   "adjustment): feature tracks -> map / image coordinates -> solvePnPRansac -> camera -> new map points, five launches per frame, no host round trip",
           "frames_per_s": round((n - 1) / dt, 1), "ms_per_frame": round(1e3 * dt / (n - 1), 4), "frames_localised": ok, "of": n - 1,
           "mean_correspondences": round(float(out["n_corr"][1:ok].mean()), 1) if ok > 1 else 0.0,
           "mean_pnp_inliers": round(float(out["n_inl"][1:ok].mean()), 1) if ok > 1 else 0.0, "map_points_at_the_end": int(out["n_map"][-1])}
    fe.ctx.close()
    return res


PNP_K = np.array([[802.832, 0, 565.427], [0, 802.832, 240.124], [0, 0, 1.0]])      # the reference's camera (test.g2o:1)


def pnp_problem(rng, n, outl, K=PNP_K):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); th = rng.uniform(0.05, 0.5)
    kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx
    t = np.array([0.3, -0.2, 30.0]) + rng.normal(0, 0.5, 3)
    X = np.concatenate([rng.uniform(-12, 12, (n, 2)), rng.uniform(-1.5, 1.5, (n, 1))], axis=1)   # ground with relief
    Xc = X @ R.T + t
    uv = ((Xc / Xc[:, 2:]) @ K.T)[:, :2] + rng.normal(0, 0.5, (n, 2))
    bad = rng.random(n) < outl
    uv[bad] += rng.uniform(-100, 100, (int(bad.sum()), 2))
    return X, uv


def pnp_pass(ctx, problems=256, points=500, outliers=0.3, steps=5, seed=11):
    """B independent cv2.solvePnPRansac problems per launch (host arrays in, poses out).  Returns (json dict, raw results)."""
    from visual_odometry_amd import _lib, geometry
    rng = np.random.default_rng(seed)
    probs = [pnp_problem(rng, points, outliers) for _ in range(problems)]
    obj = np.concatenate([p[0] for p in probs]); img = np.concatenate([p[1] for p in probs])
    off = (np.arange(problems + 1) * points).astype(np.int32)
    geometry.solve_pnp_ransac_batch(obj, img, off, PNP_K, ctx=ctx)                   # warm-up
    ctx.check(ctx.lib.vo_profile_enable(ctx.handle, 1)); ctx.check(ctx.lib.vo_profile_reset(ctx.handle))
    t0 = time.perf_counter()
    for _ in range(steps):
        status, rvec, tvec, mask, ninl = geometry.solve_pnp_ransac_batch(obj, img, off, PNP_K, ctx=ctx)
    dt = (time.perf_counter() - t0) / steps
    ms = np.zeros(_lib.VO_STAGE_COUNT, np.float32); cnt = np.zeros(_lib.VO_STAGE_COUNT, np.int32)
    ctx.check(ctx.lib.vo_profile_read(ctx.handle, ms.ctypes.data, cnt.ctypes.data))
    ctx.check(ctx.lib.vo_profile_enable(ctx.handle, 0))
    names = [ctx.lib.vo_stage_name(i).decode() for i in range(_lib.VO_STAGE_COUNT)]
    kernel_ms = float(ms[names.index("misc")] / max(cnt[names.index("misc")], 1))
    out = {"what": f"cv2.solvePnPRansac (src/visual_slam.py:231-235): {problems} problems of {points} points per launch, {int(100 * outliers)} % outliers, "
                   "100 iterations, 8 px, final pose by cv2's solvePnP(ITERATIVE); host arrays in, poses out",
           "problems_per_s": round(problems / dt, 1), "ms_per_launch_with_copies": round(1e3 * dt, 3), "kernel_ms_per_launch": round(kernel_ms, 3),
           "ok_fraction": float((status == 0).mean()), "mean_inliers": float(ninl.mean())}
    return out, dict(probs=probs, off=off, status=status, rvec=rvec, tvec=tvec, mask=mask, ninl=ninl)
