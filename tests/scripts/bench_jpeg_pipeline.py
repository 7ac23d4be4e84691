#!/usr/bin/env python3
"""Files in, relative poses out: a chunk of 257 JPEG frames (1280x720 by default) -> vo_frames_ingest_jpeg (decode + cv2.resize +
gray on the device) -> ORB detect + describe -> 256 frame pairs (match, E-RANSAC, recoverPose, triangulation), several contexts
(host threads, one HIP stream each) working on different chunks.  Only the compressed bytes cross PCIe.  Prints one JSON line.

    python tests/scripts/bench_jpeg_pipeline.py [--contexts 3] [--chunks 6] [--scale 1.0]
"""
import argparse, io, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--pairs", type=int, default=256); ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=40, help="chunks per context in the timed region")
    ap.add_argument("--scale", type=float, default=1.0, help="cv2.resize factor applied at ingest (the reference uses 0.3 on 4K footage)")
    ap.add_argument("--quality", type=int, default=90); ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--chroma", choices=["flat", "scene"], default="flat",
                    help="flat: the rendered gray view in all three channels (Cb = Cr = 128 everywhere: the hard case for the parallel entropy decoder, "
                         "whose block-in-MCU phase then synchronises late); scene: low-frequency chroma derived from the view itself, as a colour camera's files have")
    a = ap.parse_args()
    from PIL import Image
    from visual_odometry_amd import _lib, ingest, synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(a.distinct, a.width, a.height, cache_dir="/tmp", trajectory="loop")
    files = []
    for k in range(a.distinct):
        g = seq["frames"][k]
        if a.chroma == "flat": rgb = np.stack([g, g, g], -1)
        else:
            # Y = the view, Cb / Cr = smooth functions of the view's own low-pass (a property of the scene point, so the frames stay
            # geometrically consistent); JFIF YCbCr -> RGB
            gf = g.astype(np.float32)
            lp = gf
            for _ in range(4): lp = (np.roll(lp, 8, 0) + np.roll(lp, -8, 0) + np.roll(lp, 8, 1) + np.roll(lp, -8, 1) + 4 * lp) / 8
            cb = 128 + 0.35 * (lp - 128) + 20 * np.sin(lp / 17.0); cr = 128 - 0.25 * (lp - 128) + 20 * np.cos(lp / 23.0)
            rgb = np.stack([gf + 1.402 * (cr - 128), gf - 0.344136 * (cb - 128) - 0.714136 * (cr - 128), gf + 1.772 * (cb - 128)], -1).clip(0, 255).astype(np.uint8)
        b = io.BytesIO(); Image.fromarray(rgb).save(b, "JPEG", quality=a.quality, subsampling=2); files.append(b.getvalue())
    C = a.pairs
    bufs = [files[k % a.distinct] for k in range(C + 1)]
    dw, dh = int(round(a.width * a.scale)), int(round(a.height * a.scale))
    K = seq["K"].copy(); K[:2] *= a.scale
    pairs = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
    fes, packed, opts = [], [], []
    for c in range(a.contexts):
        fe = FrontEnd(dh, dw, max_frames=C + 1, max_pairs=C, nfeatures=2000, ctx=_lib.Context(0))
        pk = ingest.PackedFiles(bufs)
        fe.ingest_jpeg(pk); fe.detect(0, C + 1); r = fe.run_pairs(pairs, K, fe.make_opts())         # warm-up (allocations)
        fes.append(fe); packed.append(pk); opts.append(fe.make_opts())
    ok = [0] * a.contexts; inl = [0] * a.contexts

    def work(c):
        fe = fes[c]
        for _ in range(a.chunks):
            fe.ingest_jpeg(packed[c])
            fe.detect(0, C + 1, wait=False)
            r = fe.run_pairs(pairs, K, opts[c])
            rec = r[0] if isinstance(r, tuple) else r
            ok[c] += int((rec["status"] == 0).sum()); inl[c] += int(rec["n_inl"].sum())

    th = [threading.Thread(target=work, args=(c,)) for c in range(a.contexts)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    n = a.contexts * a.chunks * C
    print(json.dumps({"workload": f"{C + 1} JPEG files {a.width}x{a.height} (quality {a.quality}, 4:2:0, {sum(map(len, bufs)) / len(bufs) / 1024:.0f} KiB each) per chunk "
                                  f"-> decode -> resize x{a.scale} -> ORB 2000 -> {C} pairs; {a.contexts} contexts x {a.chunks} chunks; chroma: {a.chroma}",
                      "frame_pairs_per_s": round(n / dt, 1), "frames_per_s": round(a.contexts * a.chunks * (C + 1) / dt, 1), "ms_per_chunk": round(dt / (a.contexts * a.chunks) * 1e3, 2),
                      "pairs_ok_fraction": round(sum(ok) / n, 4), "mean_inliers": round(sum(inl) / max(sum(ok), 1), 1),
                      "pcie_bytes_per_pair": int(sum(map(len, bufs)) / C)}))


if __name__ == "__main__":
    main()
