#!/bin/bash
# Diagnostic (GPU box): how many pixels reach each phase of k_fast on the bench's own frames, counted by the kernel itself in a
# PRIVATE build with -DFT_STATS (global atomics; the product library is never touched).  One detection of 257 frames of the
# 1280x720 flight.  Output -> gpurun_out/fast_stats.txt (copied to profiles/r04_k_fast_survivor_statistics.txt).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/visual_odometry_amd/csrc
hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DFT_STATS -c orb_kernels.hip -o /tmp/orb_stats.o || exit 1
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_stats.so /tmp/orb_stats.o vo_api.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o -ldl || exit 1
cd $R
VO_HIP_LIBRARY=/tmp/libvo_stats.so python3 - <<'PY' | tee gpurun_out/fast_stats.txt
import ctypes, numpy as np
from visual_odometry_amd import synth, _lib
from visual_odometry_amd.frontend import FrontEnd
seq = synth.sequence(256, 1280, 720, cache_dir="/tmp", trajectory="loop")
frames = seq["frames"][np.arange(257) % 256]
fe = FrontEnd(720, 1280, max_frames=257, max_pairs=1, nfeatures=2000, nlevels=8)
fe.upload(frames)
lib = fe.ctx.lib
lib.vo_debug_fast_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
lib.vo_debug_fast_stats(None, 1)
fe.detect(0, 257)
lib.vo_debug_fast_stats(out, 0)
t, g, q, c, w, it, ov = [int(out[i]) for i in range(7)]
px = fe.stage_bytes("fast_score_nms", 257)                 # pyramid pixels of 257 frames
ring = t * 22 * 114                                          # pixels the pre-test looks at: every tile + its one-pixel ring
print(f"k_fast<false>, 257 frames of the 1280x720 flight, 8 levels, threshold 20 (counted by the kernel, -DFT_STATS build)")
print(f"  tiles                     {t:12d}   ({t // 257} per frame; {px / 257:.0f} pyramid pixels per frame)")
print(f"  pre-tested pixels         {ring:12d}   (tile + ring: {ring / px:.3f} x the pyramid pixels)")
print(f"  groups with a survivor    {g:12d}   {100.0 * g / (t * 22 * 30):6.2f} % of the groups of 4 pixels")
print(f"  pre-test survivors        {q:12d}   {100.0 * q / px:6.2f} % of the pyramid pixels, {q / t:.1f} per tile")
print(f"  FAST corners              {c:12d}   {100.0 * c / px:6.2f} %  ({100.0 * c / max(q, 1):.1f} % of the survivors)")
print(f"  listed NMS winners        {w:12d}   {100.0 * w / px:6.2f} %  (inside the 31-pixel border)")
print(f"  phase-C iterations        {it:12d}   {it / t:.2f} per tile, lanes used {100.0 * q / max(it * 128, 1):.1f} %")
print(f"  queue overflows           {ov:12d}")
PY
