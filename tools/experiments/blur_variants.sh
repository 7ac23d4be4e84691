#!/bin/bash
# Diagnostic (GPU box): k_blur_direct with different strip heights, each in its own library copy (VO_HIP_LIBRARY).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/visual_odometry_amd/csrc
for n in "$@"; do
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DBD_R=$n -c orb_kernels.hip -o /tmp/orb_v.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_b$n.so vo_api.o /tmp/orb_v.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o -ldl
  echo "== BD_R $n"
  ( cd $R; VO_HIP_LIBRARY=/tmp/libvo_b$n.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain $BENCH_ARGS | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  ',d['value'],d['ms_per_step'],d['stages']['gaussian_blur']['ms_per_launch'])" )
done
