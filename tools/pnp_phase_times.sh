#!/bin/bash
# Diagnostic (GPU box): where k_pnp_ransac spends its time — builds cut short with -DVO_PNP_STOP=n (1: after the 64 EPnP
# hypotheses of the first round, 2: after scoring + the final inlier mask, 0: full kernel incl. the LM refinement).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/visual_odometry_amd/csrc
cp ../libvo_hip.so /tmp/libvo_hip.keep
for n in 1 2 0; do
  if [ $n -eq 0 ]; then D=""; else D="-DVO_PNP_STOP=$n"; fi
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $D -c pnp_kernels.hip -o /tmp/pnp_v.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../libvo_hip.so vo_api.o orb_kernels.o match_kernels.o geom_kernels.o /tmp/pnp_v.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_kernels.o -ldl
  echo "== stop after phase $n (0 = full kernel)"
  python3 $R/tests/scripts/bench_pnp.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   kernel ms per launch', d['kernel_ms_per_launch'])"
done
cp /tmp/libvo_hip.keep ../libvo_hip.so
