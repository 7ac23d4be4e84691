import sys, io, numpy as np
sys.path.insert(0, '/root/repo')
from PIL import Image
from visual_odometry_amd import ingest, _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
for (w, h) in [(8, 8), (16, 16), (64, 48), (320, 240)]:
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(img).save(b, format="JPEG", quality=90); buf = b.getvalue()
    got = ingest.imdecode(buf, ctx=ctx) if 'ctx' in ingest.imdecode.__code__.co_varnames else ingest.imdecode(buf)
    ref = np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))[:, :, ::-1]
    d = np.abs(got.astype(int) - ref.astype(int))
    print(w, h, "max diff", d.max(), "differing px", int((d.max(axis=2) > 0).sum()), "of", w * h)
