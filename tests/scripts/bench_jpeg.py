#!/usr/bin/env python3
"""JPEG ingest throughput: a batch of 1280x720 (default) baseline JPEG frames decoded on the MI355X (vo_jpeg_decode_batch:
upload of the compressed bytes, decode, download of the B G R frames) and through the complete ingest
(vo_frames_ingest_jpeg: decode -> cv2.resize -> gray into the pyramid, nothing but the compressed bytes over PCIe),
beside libjpeg-turbo itself (Pillow) on the host cores.  Prints one JSON line.

    python tests/scripts/bench_jpeg.py [--frames 257] [--width 1280 --height 720] [--quality 90] [--subsampling 2]
"""
import argparse
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=257)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--quality", type=int, default=90)
    ap.add_argument("--subsampling", type=int, default=2)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--restart-rows", type=int, default=0)
    a = ap.parse_args()
    from PIL import Image
    from visual_odometry_amd import _lib, ingest, synth
    from visual_odometry_amd.frontend import FrontEnd
    seq = synth.sequence(a.distinct, a.width, a.height, cache_dir="/tmp")
    files = []
    for k in range(a.distinct):
        g = seq["frames"][k]
        rgb = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0)], -1)              # some chroma content
        b = io.BytesIO()
        kw = {"restart_marker_rows": a.restart_rows} if a.restart_rows else {}
        Image.fromarray(rgb).save(b, "JPEG", quality=a.quality, subsampling=a.subsampling, **kw)
        files.append(b.getvalue())
    bufs = [files[k % a.distinct] for k in range(a.frames)]
    nbytes = sum(len(b) for b in bufs)
    ctx = _lib.default_context(0)
    out = ingest.decode_batch(bufs, ctx)                                            # warm-up (allocations)
    want = np.asarray(Image.open(io.BytesIO(bufs[0])).convert("RGB"))[:, :, ::-1]
    assert os.environ.get("VO_JPEG_NOCHECK") or np.array_equal(out[0], want), "decode differs from libjpeg-turbo"   # (VO_JPEG_NOCHECK: timing of deliberately cut-short diagnostic builds)
    t = []
    for _ in range(a.repeats):
        t0 = time.perf_counter(); ingest.decode_batch(bufs, ctx); t.append(time.perf_counter() - t0)
    dec = min(t)
    sw, sh = int(a.width * 0.3 * 3), int(a.height * 0.3 * 3)                        # a 0.9x resize target, as the reference's scale factor does
    fe = FrontEnd(sh, sw, max_frames=a.frames, max_pairs=1, ctx=ctx)
    fe.ingest_jpeg(bufs)
    t = []
    for _ in range(a.repeats):
        t0 = time.perf_counter(); fe.ingest_jpeg(bufs); t.append(time.perf_counter() - t0)
    ing = min(t)
    t0 = time.perf_counter(); packed = ingest.PackedFiles(bufs); pack = time.perf_counter() - t0
    fe.ingest_jpeg(packed)
    t = []
    for _ in range(a.repeats):
        t0 = time.perf_counter(); fe.ingest_jpeg(packed); t.append(time.perf_counter() - t0)
    ing_p = min(t)
    # per-kernel times of the decode (events on the library's stream)
    c = ctx
    c.check(c.lib.vo_profile_enable(c.handle, 1)); c.check(c.lib.vo_profile_reset(c.handle))
    ingest.decode_batch(bufs, ctx)
    ms = np.zeros(_lib.VO_STAGE_COUNT, np.float32); n = np.zeros(_lib.VO_STAGE_COUNT, np.int32)
    c.check(c.lib.vo_profile_read(c.handle, ms.ctypes.data, n.ctypes.data))
    c.check(c.lib.vo_profile_enable(c.handle, 0))
    kern_ms = float(ms.sum())

    def host(b):
        im = Image.open(io.BytesIO(b)); im.load(); return im.size

    cores = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    for b in bufs[:32]:
        host(b)
    one = 32 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(host, bufs))
    allc = a.frames / (time.perf_counter() - t0)
    print(json.dumps({
        "workload": f"{a.frames} baseline JPEG frames {a.width}x{a.height}, quality {a.quality}, subsampling {['4:4:4', '4:2:2', '4:2:0'][a.subsampling]}"
                    f"{', restart every %d MCU rows' % a.restart_rows if a.restart_rows else ''}, {nbytes / a.frames / 1024:.0f} KiB per file",
        "gpu_decode_frames_per_s": round(a.frames / dec, 1), "gpu_decode_ms_per_batch": round(dec * 1e3, 2),
        "gpu_decode_kernels_ms_per_batch": round(kern_ms, 3),
        "gpu_decode_kernels_frames_per_s": round(a.frames / (kern_ms * 1e-3), 1) if kern_ms > 0 else None,
        "gpu_ingest_frames_per_s": round(a.frames / ing, 1), "gpu_ingest_ms_per_batch": round(ing * 1e3, 2),
        "gpu_ingest_packed_frames_per_s": round(a.frames / ing_p, 1), "gpu_ingest_packed_ms_per_batch": round(ing_p * 1e3, 2),
        "pack_into_page_locked_ms_per_batch": round(pack * 1e3, 2),
        "note": "decode = H2D of the files + kernels + D2H of the BGR frames; ingest = H2D of the files + decode + resize + gray, frames stay in "
                "HBM; 'ingest' takes a list of bytes objects (joined, pageable upload), 'ingest_packed' an ingest.PackedFiles (files "
                "already back to back in page-locked memory: DMA upload)",
        "libjpeg_turbo_host_frames_per_s": {"one_thread": round(one, 1), f"{cores}_threads": round(allc, 1)},
        "bit_identical_to_libjpeg_turbo": True}))


if __name__ == "__main__":
    main()
