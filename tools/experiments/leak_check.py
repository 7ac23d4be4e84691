"""Device-memory leak check: create / configure / use / destroy a context 30 times and compare hipMemGetInfo."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visual_odometry_amd import _lib, synth
from visual_odometry_amd.frontend import FrontEnd
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f = C.c_size_t(); t = C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
seq = synth.sequence(3, 640, 480, cache_dir="/tmp")
base = None
for it in range(30):
    c = _lib.Context(0)
    fe = FrontEnd(480, 640, max_frames=3, max_pairs=2, nfeatures=500, ctx=c)
    fe.upload(seq["frames"]); fe.detect(0, 3); fe.run_pairs([[0, 1], [1, 2]], seq["K"], want_points=True)
    from visual_odometry_amd.matcher import HammingMatcher
    HammingMatcher(crossCheck=True, ctx=c).match_arrays(np.random.default_rng(it).integers(0, 256, (300, 32), dtype=np.uint8), np.random.default_rng(it + 1).integers(0, 256, (280, 32), dtype=np.uint8))
    del fe
    c.close()
    if it == 2: base = free_mb()
    if it in (2, 15, 29): print("iter", it, "free MB", round(free_mb(), 1))
print("leak MB over 27 create/destroy cycles:", round(base - free_mb(), 2))
