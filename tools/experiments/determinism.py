"""Two contexts alternating chunks vs the synchronous single-context run: every chunk must be bit-identical; also
reports which stage differs if not (features / matches).  The test-suite version is
tests/test_gpu_pipeline.py::test_overlapped_contexts_are_deterministic."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import synth
from visual_odometry_amd.frontend import FrontEnd
C = 128
seq = synth.sequence(17, 1280, 720, cache_dir="/tmp")
order = [(i % 32 if i % 32 < 17 else 32 - i % 32) for i in range(C + 1)]
frames = seq["frames"][order]
pairs = np.stack([np.arange(C), np.arange(C) + 1], 1).astype(np.int32)
fes = [FrontEnd(720, 1280, max_frames=C + 1, max_pairs=C, nfeatures=2000) for _ in range(2)]
for f in fes: f.upload(frames)
opts = fes[0].make_opts(want_points=True)
ref = None; bad = 0
inflight = [None, None]
def key(r): return np.concatenate([r[k].astype(np.float64).ravel() for k in ("status","n_match","n_inl","n_good","ransac_iters","R","t")])
for it in range(int(os.environ.get("VO_DET_ITERS", "12"))):
    k = it % 2
    f = fes[k]
    if inflight[k] is not None:
        f.wait()
        v = key(inflight[k][0])
        if ref is None: ref = v.copy()
        elif not np.array_equal(ref, v):
            bad += 1; d = np.nonzero(ref != v)[0]; print("iteration", it, "differs at", len(d), "values, first", d[:5])
    f.detect(0, C + 1, wait=False, after=fes[1 - k])
    inflight[k] = f.run_pairs(pairs, seq["K"], opts, wait=False)
for k in range(2):
    fes[k].wait(); v = key(inflight[k][0])
    if not np.array_equal(ref, v): bad += 1; print("final differs", k)
print("mismatching chunks:", bad, "mean inl", ref.reshape(-1)[2*C:3*C].mean())
# after the overlapped loop: are the two contexts' features identical?  and their NN tables / matches?
nd = 0
for sl in range(C + 1):
    a, b = fes[0].features(sl), fes[1].features(sl)
    for k in ("xy", "desc", "angle", "response", "octave"):
        if a[k].shape != b[k].shape or not np.array_equal(a[k], b[k]):
            nd += 1; print("overlapped: slot", sl, k, "differs", a[k].shape, b[k].shape); break
print("slots with different features between the two contexts:", nd)
nm = 0
for p in range(C):
    a, b = fes[0].pair_matches(p), fes[1].pair_matches(p)
    if len(a[0]) != len(b[0]) or not all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])):
        nm += 1
        if nm < 5: print("pair", p, "matches differ", len(a[0]), len(b[0]))
print("pairs with different matches:", nm)
# sequential, one context, synchronous
bad2 = 0; ref2 = None
f = fes[0]
for it in range(8):
    f.detect(0, C + 1)
    r, _ = f.run_pairs(pairs, seq["K"], opts)
    v = key(r)
    if ref2 is None: ref2 = v.copy()
    elif not np.array_equal(ref2, v): bad2 += 1; d = np.nonzero(ref2 != v)[0]; print("seq iteration", it, "differs at", len(d), "first", d[:5])
print("sequential mismatching:", bad2, "equal to overlapped ref:", np.array_equal(ref, ref2))
# which stage differs: compare features of two detections
fa = [f.features(s) for s in range(0, 8)]
f.detect(0, C + 1)
fb = [f.features(s) for s in range(0, 8)]
for s in range(8):
    for k in ("xy", "desc", "angle", "response"):
        if not np.array_equal(fa[s][k], fb[s][k]): print("slot", s, k, "differs")
