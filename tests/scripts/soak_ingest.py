#!/usr/bin/env python3
"""Checker script (GPU box): cv2.resize as the reference calls it (INTER_LINEAR to a scaled size, visual_slam.py:346-352; INTER_AREA
shrinking, image_and_keypoints.py:42) on random source / destination sizes, 1 / 3 / 4 channels, and the batched frame ingest
(resize -> gray -> level 0, checked through the resized frames it returns): ingest.resize and FrontEnd.ingest against the oracle.
    python tests/scripts/soak_ingest.py [--seconds 200] [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=200); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from visual_odometry_amd import _lib, ingest
    from visual_odometry_amd.frontend import FrontEnd
    from oracle import oracle as O
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    t0 = tick = time.time(); nlin = narea = nbatch = 0
    while time.time() - t0 < a.seconds:
        if time.time() - tick > 30: tick = time.time(); print(f"... linear {nlin}, area {narea}, batched ingests {nbatch} identical so far", flush=True)
        sh, sw = int(rng.integers(1, 500)), int(rng.integers(1, 700))
        cn = int(rng.choice([1, 3, 4]))
        src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
        if rng.random() < 0.3: src = (src // 64 * 85).astype(np.uint8)               # few levels: many exact .5 roundings
        mode = rng.random()
        if mode < 0.25: dw, dh = max(1, sw // 2), max(1, sh // 2)                      # the exact 2x shrink (own code path)
        elif mode < 0.5: f = float(rng.choice([0.3, 0.5, 0.9, 1.0, 1.5, 2.0])); dw, dh = max(1, int(sw * f)), max(1, int(sh * f))
        else: dw, dh = int(rng.integers(1, 800)), int(rng.integers(1, 600))
        tag = dict(sh=sh, sw=sw, cn=cn, dw=dw, dh=dh, seed=a.seed)
        if not np.array_equal(ingest.resize(src, (dw, dh), ctx=ctx), O.resize_linear(src, dw, dh)): print("MISMATCH linear", tag); sys.exit(1)
        nlin += 1
        if dw <= sw and dh <= sh:
            if not np.array_equal(ingest.resize(src, (dw, dh), interpolation=ingest.INTER_AREA, ctx=ctx), O.resize_area(src, dw, dh)): print("MISMATCH area", tag); sys.exit(1)
            narea += 1
        if rng.random() < 0.1 and dw >= 32 and dh >= 32 and cn != 4 and sh > 4 and sw > 4:   # ([F, H, W] with W <= 4 reads as one colour image)
            F = int(rng.integers(1, 4))
            frames = rng.integers(0, 256, (F, sh, sw) if cn == 1 else (F, sh, sw, cn), dtype=np.uint8)
            fe = FrontEnd(dh, dw, max_frames=F, max_pairs=1, nfeatures=100, ctx=ctx)
            got = fe.ingest(frames, want_resized=True)
            for k in range(F):
                if not np.array_equal(got[k], O.resize_linear(frames[k], dw, dh)): print("MISMATCH batched ingest", tag, k); sys.exit(1)
            fe.detect(0, F)
            f0 = fe.features(0); want = O.resize_linear(frames[0], dw, dh)
            o = O.orb_detect_and_compute(want, O.orb_params(nfeatures=100))
            if not f0["truncated"] and not (np.array_equal(f0["xy"], o["xy"]) and np.array_equal(f0["desc"], o["desc"])): print("MISMATCH ingest -> ORB", tag); sys.exit(1)
            nbatch += 1
    print(json.dumps({"resize_linear": nlin, "resize_area": narea, "batched_ingests": nbatch, "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
