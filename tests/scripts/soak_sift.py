#!/usr/bin/env python3
"""Checker script (GPU box): random images (size, texture class) and SIFT parameters (layers, thresholds, sigma, nfeatures) through
the batched detector — single frames and small batches through FrontEnd(detector="sift") — and the oracle: keypoints and
descriptors bit for bit; the L2 matcher (int8 matrix-core path) against the oracle's float batchDistance order on the descriptors.
    python tests/scripts/soak_sift.py [--seconds 300] [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def texture(rng, h, w):
    kind = int(rng.integers(0, 4))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        b = int(rng.integers(2, 9))
        img = np.kron(rng.integers(0, 256, (h // b + 1, w // b + 1)), np.ones((b, b)))[:h, :w] + rng.normal(0, 6, (h, w))
    elif kind == 1:
        img = 40.0 + 0.4 * xx + 0.2 * yy
        for _ in range(int(rng.integers(5, 40))):
            cy, cx, sg = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(1.5, 9)
            img = img + rng.uniform(-90, 90) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
    elif kind == 2:
        img = rng.integers(0, 256, (h, w)).astype(np.float64)
    else:
        img = np.full((h, w), float(rng.integers(0, 256)))
        for _ in range(int(rng.integers(1, 14))):
            y, x = int(rng.integers(0, max(h - 8, 1))), int(rng.integers(0, max(w - 8, 1)))
            img[y:y + int(rng.integers(4, 50)), x:x + int(rng.integers(4, 50))] = float(rng.integers(0, 256))
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=300); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from visual_odometry_amd import _lib
    from visual_odometry_amd.detector import SiftDetector
    from visual_odometry_amd.matcher import L2Matcher
    from oracle import oracle as O
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    t0 = tick = time.time(); n = nkp = nm = ntr = 0
    while time.time() - t0 < a.seconds:
        if time.time() - tick > 30: tick = time.time(); print(f"... {n} images ({nkp} keypoints), {nm} match sets identical so far", flush=True)
        h, w = int(rng.integers(16, 300)), int(rng.integers(16, 360))
        img = texture(rng, h, w)
        kw = dict(nOctaveLayers=int(rng.integers(2, 6)), contrastThreshold=float(rng.uniform(0.01, 0.08)), edgeThreshold=float(rng.uniform(4, 16)),
                  sigma=float(rng.uniform(1.0, 2.1)),          # (up to 63 taps are built: two or more layers per octave at these sigmas)
                  nfeatures=int(rng.choice([0, 0, 0, 50, 400])))
        want = O.sift_detect_and_compute(img, nfeatures=kw["nfeatures"], n_layers=kw["nOctaveLayers"], contrast_threshold=kw["contrastThreshold"],
                                         edge_threshold=kw["edgeThreshold"], sigma=kw["sigma"])
        got = SiftDetector(ctx=ctx, **kw).detect_arrays(img)
        tag = (h, w, kw, a.seed, n)
        if got.get("truncated"): ntr += 1; continue             # a list capacity was reached: flagged (RuntimeWarning), not claimed exact
        if len(got["xy"]) != want["n_found"]:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_sift_fail.npz"), img=img, **{k: np.array(v) for k, v in kw.items()})
            print("MISMATCH count", tag, len(got["xy"]), want["n_found"], "truncated" if got.get("truncated") else ""); sys.exit(1)
        for key in ("xy", "size", "angle", "response", "octave", "desc"):
            if not np.array_equal(got[key], want[key]): print("MISMATCH", key, tag); sys.exit(1)
        n += 1; nkp += want["n_found"]
        d1 = want["desc"]
        if len(d1) >= 8:
            d2 = d1[rng.permutation(len(d1))[: max(4, len(d1) * 3 // 4)]].copy()
            d2 = np.clip(d2 + rng.integers(-2, 3, d2.shape) * (rng.random(d2.shape) < 0.1), 0, 255).astype(np.float32)
            d2 = np.concatenate([d2, d2[: len(d2) // 5]])
            rq, rt, rd = O.match_l2(d1, d2, 2)
            ms = L2Matcher(crossCheck=True, ctx=ctx).match(d1, d2)
            if not (np.array_equal([m.queryIdx for m in ms], rq) and np.array_equal([m.trainIdx for m in ms], rt) and np.array_equal(np.array([m.distance for m in ms], np.float32), rd)):
                print("MISMATCH matcher", tag); sys.exit(1)
            nm += 1
    print(json.dumps({"images": n, "keypoints": int(nkp), "match_sets": nm, "capacity_flagged": ntr, "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
