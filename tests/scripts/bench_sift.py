#!/usr/bin/env python3
"""SIFT throughput on one MI355X (per-image call: upload, pyramids, extrema, refinement, descriptors, download) beside the
scalar CPU oracle.  Prints one JSON line.   python tests/scripts/bench_sift.py [--width 1280 --height 720] [--frames 8]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--oracle-frames", type=int, default=1)
    ap.add_argument("--contexts", default="2,4,8", help="also measure N contexts (N host threads, N HIP streams) working on different frames")
    a = ap.parse_args()
    from visual_odometry_amd import _lib, synth
    from visual_odometry_amd.detector import SiftDetector
    seq = synth.sequence(a.frames, a.width, a.height, cache_dir="/tmp")
    ctx = _lib.default_context(0)
    det = SiftDetector(ctx=ctx)
    r = det.detect_arrays(seq["frames"][0])
    t0 = time.perf_counter()
    n = 0
    for k in range(a.frames):
        n += len(det.detect_arrays(seq["frames"][k])["xy"])
    dt = (time.perf_counter() - t0) / a.frames
    out = {"workload": f"SIFT detectAndCompute, {a.width}x{a.height} synthetic drone frames, cv2 defaults", "keypoints_per_frame": n // a.frames,
           "gpu_ms_per_frame": round(dt * 1e3, 2), "gpu_frames_per_s": round(1 / dt, 1)}
    # several contexts side by side: the per-image call is a chain of ~200 small kernels and two host synchronisations, so one
    # stream leaves most of the GPU idle; independent frames on independent contexts fill it (ctypes releases the GIL)
    import threading
    multi = {}
    for nc in [int(x) for x in a.contexts.split(",") if x]:
        dets = [SiftDetector(ctx=_lib.Context(0)) for _ in range(nc)]
        for d in dets:
            d.detect_arrays(seq["frames"][0])
        per = max(a.frames, 8)

        def work(d, j):
            for k in range(per):
                d.detect_arrays(seq["frames"][(j + k) % a.frames])

        th = [threading.Thread(target=work, args=(d, j)) for j, d in enumerate(dets)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        multi[str(nc)] = round(nc * per / (time.perf_counter() - t0), 1)
        del dets
    out["gpu_frames_per_s_with_n_contexts"] = multi
    if a.oracle_frames:
        from oracle import oracle as O
        t0 = time.perf_counter()
        ro = O.sift_detect_and_compute(seq["frames"][0])
        out["oracle_one_thread_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["identical_to_oracle"] = bool(np.array_equal(ro["desc"], r["desc"]) and np.array_equal(ro["xy"], r["xy"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
