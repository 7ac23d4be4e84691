#!/bin/bash
# Diagnostic (GPU box): end-to-end A/B on ONE box of several versions of orb_kernels.hip (copies named orb_kernels_<tag>.hip next
# to the source; not committed) against the current one — box-to-box variation (+-3 %) hides anything smaller.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/visual_odometry_amd/csrc
tags=""
for f in orb_kernels_*.hip; do
  t=${f#orb_kernels_}; t=${t%.hip}; tags="$tags $t"
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -c $f -o /tmp/orb_$t.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_$t.so vo_api.o /tmp/orb_$t.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o -ldl
done
cd $R
for i in 1 2 3; do
  for v in $tags current; do
    if [ $v = current ]; then unset VO_HIP_LIBRARY; else export VO_HIP_LIBRARY=/tmp/libvo_$v.so; fi
    python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --steps 200 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v',d['value'],d['ms_per_step'])"
  done
done
