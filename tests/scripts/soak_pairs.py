#!/usr/bin/env python3
"""Checker script (GPU box): the WHOLE per-pair path, batched and device resident (frames -> ORB in cv2's order -> Hamming cross-check
-> findEssentialMat -> recoverPose -> triangulation), in the `opencv300` root-finder mode against oracle.pair on rendered flights of
random size, step, yaw, feature count and pair stride: counts, E, R | t bit for bit, points to 1e-9 in direction.
    python tests/scripts/soak_pairs.py [--seconds 300] [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=300); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from visual_odometry_amd import _lib, synth
    from visual_odometry_amd.frontend import FrontEnd
    from oracle import oracle as O
    rng = np.random.default_rng(a.seed)
    t0 = tick = time.time(); npairs = nfail_both = 0
    while time.time() - t0 < a.seconds:
        if time.time() - tick > 30: tick = time.time(); print(f"... {npairs} pairs identical so far", flush=True)
        w, h = int(rng.integers(20, 90)) * 8, int(rng.integers(15, 60)) * 8
        nf = int(rng.choice([200, 500, 1000, 2000])); nfr = int(rng.integers(3, 6))
        seq = synth.sequence(nfr, w, h, step=float(rng.uniform(0.3, 2.5)), yaw_deg=float(rng.uniform(0, 2.0)), seed=int(rng.integers(0, 1 << 30)), workers=8)
        frames, K = seq["frames"], seq["K"]
        ctx = _lib.Context(0); ctx.set_poly_solver("opencv300")
        fe = FrontEnd(h, w, max_frames=nfr, max_pairs=2 * nfr, nfeatures=nf, ctx=ctx)
        fe.upload(frames); fe.detect(0, nfr)
        pairs = [[i, i + 1] for i in range(nfr - 1)] + [[0, nfr - 1]]
        res, X = fe.run_pairs(pairs, K, fe.make_opts(want_points=True))
        p = O.orb_params(nfeatures=nf)
        for i, (f1, f2) in enumerate(pairs):
            ref = O.pair(frames[f1], frames[f2], p, K)
            tag = dict(w=w, h=h, nf=nf, pair=(f1, f2), seed=a.seed, n=npairs)
            if (res[i]["status"] != 0) != (ref["rc"] != 0): print("MISMATCH verdict", tag, res[i]["status"], ref["rc"]); sys.exit(1)
            if ref["rc"] != 0: nfail_both += 1; continue
            got = (res[i]["n_kp1"], res[i]["n_kp2"], res[i]["n_match"], res[i]["n_inl"], res[i]["n_good"])
            want = (ref["n_kp1"], ref["n_kp2"], ref["n_match"], ref["n_inl"], ref["n_good"])
            if got != want: print("MISMATCH counts", tag, got, want); sys.exit(1)
            if not (np.array_equal(res[i]["R"].reshape(3, 3), ref["R"]) and np.array_equal(res[i]["t"].reshape(3, 1), ref["t"]) and np.array_equal(res[i]["E"].reshape(3, 3), ref["E"])):
                print("MISMATCH E / R / t", tag); sys.exit(1)
            n = ref["n_inl"]
            Xg, Xr = X[i][:, :n], ref["X"][:, :n]
            hg, hr = Xg / np.linalg.norm(Xg, axis=0), Xr / np.linalg.norm(Xr, axis=0)
            if n and (1.0 - np.abs((hg * hr).sum(axis=0))).max() > 1e-9: print("MISMATCH points", tag); sys.exit(1)
            npairs += 1
        ctx.close()
    print(json.dumps({"pairs": npairs, "pairs_rejected_by_both": nfail_both, "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
