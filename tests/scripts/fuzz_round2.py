#!/usr/bin/env python3
"""Extended differential fuzz of the round-2 kernels against the oracle (run on the GPU box; minutes, not part of the test
suite): the FP16 three-input cornerScore over many image classes / thresholds, the FP4 matrix-core matcher over many descriptor
sets with planted ties, the RANSAC round logic over many two-view problems.  Prints one line per family and exits non-zero on
the first difference."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from visual_odometry_amd import _lib
from visual_odometry_amd.detector import OrbDetector
from visual_odometry_amd.matcher import HammingMatcher
from visual_odometry_amd import geometry
import twoview

N = int(os.environ.get("FUZZ_N", "150"))
ctx = _lib.default_context(0)
rng = np.random.default_rng(2026)


def image(kind, h, w):
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 1:                                                          # low contrast around the threshold
        return (128 + rng.integers(-24, 25, (h, w))).astype(np.uint8)
    if kind == 2:                                                          # blocks with hard edges, saturated
        b = rng.integers(0, 2, (h // 5 + 1, w // 5 + 1)) * 255
        return np.kron(b, np.ones((5, 5)))[:h, :w].astype(np.uint8)
    if kind == 3:                                                          # smooth gradients + sparse salt
        yy, xx = np.mgrid[0:h, 0:w]
        a = (xx * 3 + yy * 2) % 256
        m = rng.random((h, w)) < 0.02
        a[m] = rng.integers(0, 256, int(m.sum()))
        return a.astype(np.uint8)
    g = rng.integers(0, 256, (h // 3 + 1, w // 3 + 1)).astype(np.float32)
    g = np.kron(g, np.ones((3, 3), np.float32))[:h, :w] + rng.normal(0, 20, (h, w))
    return np.clip(g, 0, 255).astype(np.uint8)


t0 = time.time()
for it in range(N):
    h, w = int(rng.integers(70, 400)), int(rng.integers(70, 500))
    thr = int(rng.choice([5, 10, 20, 20, 20, 35, 60]))
    img = image(it % 5, h, w)
    nl = int(rng.integers(1, 5))
    det = OrbDetector(nfeatures=300, nlevels=nl, fastThreshold=thr)
    p = O.orb_params(nfeatures=300, nlevels=nl, fast_threshold=thr)
    lw, lh = O.level_geometry(h, w, p)[:2]
    got = det.stage_levels("vo_stage_fast_scores", img, [(int(a), int(b)) for a, b in zip(lw, lh)])
    sizes = [(int(a), int(b)) for a, b in zip(lw, lh)]
    pyr = O.pyramid(img, p)
    for l, lvl in enumerate(pyr):
        ref = O.fast_score_nms(lvl, thr)
        if not np.array_equal(got[l], ref):
            print("FAST score differs", it, h, w, thr, l, np.count_nonzero(got[l] != ref)); sys.exit(1)
    if it % 2 == 0:                                                         # pyramid levels and their 7x7 blur: every width residue
        gp, gb = det.stage_levels("vo_stage_pyramid", img, sizes), det.stage_levels("vo_stage_blur", img, sizes)
        for l, lvl in enumerate(pyr):
            if not np.array_equal(gp[l], lvl) or not np.array_equal(gb[l], O.gaussian_blur7(lvl)):
                print("pyramid / blur differs", it, h, w, l); sys.exit(1)
print(f"FAST cornerScore (+ pyramid, blur on every second image): {N} images identical ({time.time() - t0:.0f} s)", flush=True)

t0 = time.time()
for it in range(N):
    nq, nt = int(rng.integers(1, 2300)), int(rng.integers(1, 2300))
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    k = min(nq, nt)
    sel = rng.permutation(nt)[:k]
    q[:k] = t[sel]                                                          # planted near-duplicates and exact ties
    flips = rng.integers(0, 12, k)
    for i in np.nonzero(flips)[0][:400]:
        for b in rng.integers(0, 256, flips[i]): q[i, b // 8] ^= 1 << (b % 8)
    if it % 7 == 0 and nt > 3: t[1] = t[0]; t[nt - 1] = t[0]                # duplicated train rows: index tie-break
    for mode in (0, 1, 2):
        m = HammingMatcher(crossCheck=mode > 0, legacy_crosscheck=mode == 1)
        a, b = m.match_arrays(q, t), O.match_hamming(q, t, mode)
        if not all(np.array_equal(x, y) for x, y in zip(a, b)):
            print("matcher differs", it, nq, nt, mode); sys.exit(1)
    a, b = HammingMatcher().ratio_match_arrays(q, t, 0.8), O.knn2_ratio_hamming(q, t, 0.8)
    if not all(np.array_equal(x, y) for x, y in zip(a, b)):
        print("ratio matcher differs", it, nq, nt); sys.exit(1)
print(f"FP4 matcher: {N} descriptor sets x 4 modes identical ({time.time() - t0:.0f} s)", flush=True)

t0 = time.time()
O.set_dk_early_exit(True)                                                   # the product's default root-finder rule
for it in range(max(N // 3, 20)):
    pr = twoview.fuzz_problem(rng)
    E, mask = geometry.findEssentialMat(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], threshold=pr["thresh"])
    rc, Eo, mo, ninl = O.find_essential_ransac(pr["p1"], pr["p2"], pr["K"], prob=pr["prob"], thresh=pr["thresh"])
    if (E is None) != (rc != 0) or (E is not None and (not np.array_equal(mask.ravel(), mo) or not np.array_equal(E, Eo[0]))):
        print("RANSAC differs", it, pr["tag"]); sys.exit(1)
O.set_dk_early_exit(False)
print(f"E-RANSAC: {max(N // 3, 20)} two-view problems identical ({time.time() - t0:.0f} s)", flush=True)
