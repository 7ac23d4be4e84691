#!/bin/bash
# Diagnostic (GPU box): k_fast built with extra macros, each variant in its own library copy.   tools/fast_flag_variants.sh "-DX" "-DY -DZ" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/visual_odometry_amd/csrc
i=0
for fl in "$@"; do
  i=$((i+1))
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $fl -c orb_kernels.hip -o /tmp/orb_f.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_f$i.so vo_api.o /tmp/orb_f.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o -ldl
  echo "== flags: $fl"
  ( cd $R; VO_HIP_LIBRARY=/tmp/libvo_f$i.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain $BENCH_ARGS | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  ',d['value'],d['ms_per_step'],d['stages']['fast_score_nms']['ms_per_launch'])" )
done
