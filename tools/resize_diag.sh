#!/bin/bash
# Experiment (GPU box): k_resize_strip with 1 / 2 / 4 wavefronts per workgroup (tile heights 16 / 32 / 64 rows).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/visual_odometry_amd/csrc
cp ../libvo_hip.so /tmp/libvo_hip.keep
for n in 4 2 1; do
  F="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DRS2_WAVES=$n"
  hipcc $F -c orb_kernels.hip -o /tmp/orb_v.o && hipcc $F -c vo_api.hip -o /tmp/api_v.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../libvo_hip.so /tmp/api_v.o /tmp/orb_v.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o -ldl
  (cd $R && python3 -m pytest tests/test_gpu_orb.py -q -x 2>&1 | tail -1; python3 bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('waves $n pyramid', d['stages']['pyramid_resize'])")
done
cp /tmp/libvo_hip.keep ../libvo_hip.so
