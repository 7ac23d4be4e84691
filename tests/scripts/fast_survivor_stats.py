#!/usr/bin/env python3
"""What fraction of the pixels reaches each phase of k_fast, and what alternative pre-tests / NMS-aware schemes would change —
measured on the bench's own frames (the seeded 1280x720 flight, all 8 pyramid levels), on the CPU in numpy.  Test
infrastructure (uses the oracle's pyramid); writes the table that profiles/r04_k_fast_survivor_statistics.txt quotes.

  compass     k_fast phase B: two adjacent compass pixels (ring 0, 4, 8, 12) both brighter than v + t or both darker than v - t
  pairs4      OpenCV's own early-out (fast.cpp): one pixel of each antipodal pair (0,8) (2,10) (4,12) (6,14) beyond the threshold, same polarity
  pairs8      all eight antipodal pairs
  corner      FAST-9/16 corners (score > 0)
  winner      3x3 NMS winners
  scheme (b)  exact scores only where the NMS outcome needs them, using the compass upper bound U = max(X - v, v - Y) phase B already has
"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/scripts/ -> repo root
sys.path.insert(0, ROOT)
from oracle import oracle as O                       # noqa: E402
from visual_odometry_amd import synth                # noqa: E402

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def ring_stack(img):
    h, w = img.shape
    v = img[3:h - 3, 3:w - 3].astype(np.int16)
    r = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int16) for dx, dy in RING])
    return v, r


def stats(img, t=20):
    v, r = ring_stack(img)
    d = r - v[None]                                   # ring - v
    br, dk = d > t, d < -t
    def pairs(ks):
        b = np.ones_like(v, bool); k_ = np.ones_like(v, bool)
        for k in ks:
            b &= br[k] | br[k + 8]; k_ &= dk[k] | dk[k + 8]
        return b | k_
    compass = pairs([0, 4])
    p4 = pairs([0, 2, 4, 6]); p8 = pairs(range(8))
    # exact score: max over 9-arcs of min(ring - v) / min(v - ring)
    dd = np.concatenate([d, d[:8]])
    amin = np.full_like(v, -999); bmin = np.full_like(v, -999)
    for k in range(16):
        w9 = dd[k:k + 9]
        amin = np.maximum(amin, (-w9).min(0)); bmin = np.maximum(bmin, w9.min(0))
    m = np.maximum(amin, bmin)
    score = np.where(m > t, m - 1, 0).astype(np.int16)
    corner = score > 0
    assert not (corner & ~compass).any() and not (corner & ~p8).any()
    # compass upper bound on max(A, B)
    X = np.minimum(np.maximum(r[0], r[8]), np.maximum(r[4], r[12])); Y = np.maximum(np.minimum(r[0], r[8]), np.minimum(r[4], r[12]))
    U = np.maximum(X - v, v - Y)
    assert (m[compass] <= U[compass]).all()
    U8 = np.minimum.reduce([np.maximum(d[k], d[k + 8]) for k in range(8)]); L8 = np.minimum.reduce([np.maximum(-d[k], -d[k + 8]) for k in range(8)])
    U8 = np.maximum(U8, L8)
    def nbr_max(a):
        p = np.pad(a, 1, constant_values=-1)
        return np.maximum.reduce([p[1 + dy:p.shape[0] - 1 + dy, 1 + dx:p.shape[1] - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)])
    winner = corner & (score > nbr_max(score))
    out = dict(px=v.size, compass=compass.sum(), pairs4=p4.sum(), pairs8=p8.sum(), corner=corner.sum(), winner=winner.sum())
    # scheme (b): round 1 = candidates whose bound is a local maximum among candidates; a candidate p with a scored neighbour q,
    # s(q) >= U(p) - 1, cannot win; the rest is scored in round 2; round 3 = killed candidates whose exact score a would-be winner needs
    for name, cand, UB in (("b_compass", compass, U), ("b_pairs8", p8, U8)):
        Uc = np.where(cand, UB, -1)
        r1 = cand & (Uc >= nbr_max(Uc))
        s1 = np.where(r1, score, -1)
        killed = cand & ~r1 & (nbr_max(s1) >= UB - 1)
        r2 = cand & ~r1 & ~killed
        known = np.where(r1 | r2, score, -1)
        would_win = (r1 | r2) & corner & (score > nbr_max(known))
        ub_killed = np.where(killed, UB - 1, -1)
        # a would-be winner r is ambiguous if a killed neighbour's bound reaches its score
        amb = would_win & (nbr_max(ub_killed) >= score)
        pa = np.pad(amb, 1)
        near_amb = np.logical_or.reduce([pa[1 + dy:pa.shape[0] - 1 + dy, 1 + dx:pa.shape[1] - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
        r3 = killed & near_amb & (UB - 1 >= 0)
        out[name] = int(r1.sum() + r2.sum() + r3.sum())
        out[name + "_r1"] = int(r1.sum()); out[name + "_r2"] = int(r2.sum()); out[name + "_r3"] = int(r3.sum())
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    seq = synth.sequence(256, 1280, 720, cache_dir="/tmp", trajectory="loop")
    p = O.orb_params(nfeatures=2000, nlevels=8)
    tot = {}
    for f in np.linspace(0, 255, n).astype(int):
        for lvl in O.pyramid(seq["frames"][f], p):
            for k, v in stats(lvl).items():
                tot[k] = tot.get(k, 0) + int(v)
    px = tot["px"]
    print(f"{n} frames of the 1280x720 flight, 8 levels, threshold 20: {px} pixels (3-pixel frame excluded)")
    for k in ("compass", "pairs4", "pairs8", "corner", "winner"):
        print(f"  {k:10s} {tot[k]:10d}  {100.0 * tot[k] / px:6.2f} % of the pixels")
    for k in ("b_compass", "b_pairs8"):
        base = tot["compass"] if k == "b_compass" else tot["pairs8"]
        print(f"  scheme {k}: exact scores needed {tot[k]} = {100.0 * tot[k] / base:.1f} % of its candidates ({100.0 * tot[k] / px:.2f} % of the pixels); "
              f"rounds {tot[k + '_r1']} / {tot[k + '_r2']} / {tot[k + '_r3']}")


if __name__ == "__main__":
    main()
