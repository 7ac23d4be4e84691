"""Diagnostic (GPU box): decode a few random images of growing size (optionally with restart intervals) and report how many pixels
differ from libjpeg-turbo (Pillow) — a quick bisecting aid while changing the entropy kernels."""
import sys, io, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from PIL import Image
from visual_odometry_amd import ingest, _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
for kw in ({}, dict(restart_marker_blocks=1), dict(restart_marker_blocks=3)):
    for (w, h) in [(8, 8), (16, 16), (64, 48), (320, 240)]:
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=90, **kw); buf = b.getvalue()
        got = ingest.imdecode(buf, ctx)
        ref = np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))[:, :, ::-1]
        d = np.abs(got.astype(int) - ref.astype(int)).max(axis=2)
        ys, xs = np.nonzero(d)
        print(kw, w, h, "max diff", d.max(), "differing px", len(ys), "of", w * h, "first at", (ys[0], xs[0]) if len(ys) else None)
