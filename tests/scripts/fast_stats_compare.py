#!/usr/bin/env python3
"""Diagnostic (GPU box, after tools/experiments/fast_stats.sh built /tmp/libvo_stats.so): the kernel's own phase counters against the
same quantities computed in numpy, level by level, on frame 0 of the bench flight.  Run with VO_HIP_LIBRARY=/tmp/libvo_stats.so."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "scripts"))
from oracle import oracle as O                                    # noqa: E402
from visual_odometry_amd import synth                             # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                 # noqa: E402
from fast_survivor_stats import stats                             # noqa: E402

seq = synth.sequence(256, 1280, 720, cache_dir="/tmp", trajectory="loop")
p = O.orb_params(nfeatures=2000, nlevels=8)
for l, img in enumerate(O.pyramid(seq["frames"][0], p)):
    h, w = img.shape
    fe = FrontEnd(h, w, max_frames=1, max_pairs=1, nfeatures=2000, nlevels=1)
    lib = fe.ctx.lib
    lib.vo_debug_fast_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fe.upload(np.ascontiguousarray(img)[None])
    lib.vo_debug_fast_stats(None, 1)
    fe.detect(0, 1)
    out = (ctypes.c_ulonglong * 8)()
    lib.vo_debug_fast_stats(out, 0)
    s = stats(img)
    print(f"level {l} {w}x{h}: kernel tiles {out[0]} groups {out[1]} survivors {out[2]} corners {out[3]} winners {out[4]} overflow {out[6]} | "
          f"numpy compass {s['compass']} corners {s['corner']} winners {s['winner']} (3-px frame excluded)  ratio survivors {out[2] / max(s['compass'], 1):.2f}")
    fe.ctx.close()
