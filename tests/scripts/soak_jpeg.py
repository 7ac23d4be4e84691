#!/usr/bin/env python3
"""Checker script (GPU box): random baseline JPEG files (size, content, quality, subsampling, restart intervals, optimised tables,
grayscale, truncation) through the HIP decoder, the oracle and libjpeg-turbo (Pillow), one at a time and in mixed batches.
    python tests/scripts/soak_jpeg.py [--seconds 240] [--seed 1]
Prints one JSON line; exits 1 on the first difference (after printing the parameters that produced it)."""
import argparse, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=240); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from PIL import Image, ImageFile
    from visual_odometry_amd import _lib, ingest
    from oracle import oracle as O
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(a.seed)
    t0 = time.time(); n = 0; nbatch = 0; ntrunc = 0; ningest = 0
    pending = {}; tick = t0
    while time.time() - t0 < a.seconds:
        if time.time() - tick > 30: tick = time.time(); print(f"... {n} files identical so far", flush=True)
        h = int(rng.choice([1, 7, 8, 16, 17, 33, 64, 100, 240, 481, 720])) if rng.random() < 0.5 else int(rng.integers(1, 500))
        w = int(rng.choice([1, 8, 15, 16, 31, 64, 129, 320, 641, 1280])) if rng.random() < 0.5 else int(rng.integers(1, 700))
        kind = rng.choice(["noise", "smooth", "boxes", "flat", "saturated"])
        if kind == "noise": img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif kind == "flat": img = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
        elif kind == "saturated": img = (rng.integers(0, 2, (h, w, 3)) * 255).astype(np.uint8)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 3 + xx) % 256, (xx * 2) % 256, (yy + xx * 5) % 256], -1).astype(np.uint8)
        else:
            img = np.zeros((h, w, 3), np.uint8)
            for _ in range(20):
                y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                img[y0:y0 + int(rng.integers(1, 60)), x0:x0 + int(rng.integers(1, 60))] = rng.integers(0, 256, 3)
            img = (img.astype(np.int32) + rng.integers(-6, 7, img.shape)).clip(0, 255).astype(np.uint8)
        gray = rng.random() < 0.15
        kw = dict(quality=int(rng.choice([1, 10, 35, 50, 75, 90, 95, 100])), subsampling=int(rng.integers(0, 3)))
        if rng.random() < 0.3: kw["optimize"] = True
        r = rng.random()
        if r < 0.2: kw["restart_marker_blocks"] = int(rng.integers(1, 12))
        elif r < 0.3: kw["restart_marker_rows"] = int(rng.integers(1, 4))
        b = io.BytesIO()
        try:
            (Image.fromarray(img[:, :, 0]) if gray else Image.fromarray(img)).save(b, "JPEG", **({k: v for k, v in kw.items() if k != "subsampling"} if gray else kw))
        except OSError:                                        # (Pillow's encoder buffer is too small for some tiny noisy images)
            continue
        buf = b.getvalue()
        trunc = rng.random() < 0.1 and len(buf) > 700
        if trunc:
            buf = buf[:int(len(buf) * rng.uniform(0.4, 0.98))] + b"\xff\xd9"; ntrunc += 1
        try: got = ingest.imdecode(buf, ctx)
        except (_lib.VoError, NotImplementedError): got = None
        try: want = O.jpeg_decode(buf)
        except ValueError: want = None
        if got is None or want is None:                        # (a cut inside the headers: both must refuse the file)
            if (got is None) != (want is None):
                print("ONE SIDE REFUSED", dict(h=h, w=w, kind=str(kind), gray=bool(gray), trunc=bool(trunc), seed=a.seed, n=n, hip=got is not None, **kw)); sys.exit(1)
            continue
        ok = np.array_equal(got, want)
        if ok and not trunc:
            ok = np.array_equal(got, np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))[:, :, ::-1])
        if not ok:
            print("MISMATCH", dict(h=h, w=w, kind=str(kind), gray=bool(gray), trunc=bool(trunc), seed=a.seed, n=n, **kw)); sys.exit(1)
        n += 1
        if h >= 70 and w >= 70 and not trunc and rng.random() < 0.04:
            # the frame ingest at the file's own size: gray level 0 straight from the decoder's colour conversion, then ORB
            from visual_odometry_amd.frontend import FrontEnd
            fe = FrontEnd(h, w, max_frames=1, max_pairs=1, nfeatures=200, ctx=ctx)
            fe.ingest_jpeg([buf]); fe.detect(0, 1)
            f, o = fe.features(0), O.orb_detect_and_compute(want, O.orb_params(nfeatures=200))
            if f["truncated"]: pass                            # (a capacity of DESIGN.md section 7 was reached — saturated noise: flagged, not claimed exact)
            elif not (np.array_equal(f["desc"], o["desc"]) and np.array_equal(f["xy"], o["xy"])):
                print("INGEST MISMATCH", dict(h=h, w=w, kind=str(kind), gray=bool(gray), seed=a.seed, n=n, **kw)); sys.exit(1)
            ningest += 1
        pending.setdefault((h, w), []).append((buf, got))
        if len(pending[(h, w)]) >= 3 or (rng.random() < 0.02 and pending):
            key = (h, w) if len(pending[(h, w)]) >= 3 else next(iter(pending))
            items = pending.pop(key)
            outs = ingest.decode_batch([x[0] for x in items], ctx)              # files with DIFFERENT tables / restart settings in one launch
            for o, (_, g) in zip(outs, items):
                if not np.array_equal(o, g): print("BATCH MISMATCH", key, a.seed, n); sys.exit(1)
            nbatch += 1
    print(json.dumps({"files": n, "truncated": ntrunc, "mixed_batches": nbatch, "ingests_checked_through_orb": ningest, "seconds": round(time.time() - t0, 1), "identical": True}))


if __name__ == "__main__":
    main()
