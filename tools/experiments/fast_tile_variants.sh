#!/bin/bash
# Diagnostic (GPU box): k_fast with other tile heights / widths / queue capacities (LDS per wave decides the waves per SIMD).  The units
# that see the tile geometry (vo_api, orb_kernels, cv2order_kernels) are rebuilt with the same macros into a private library copy.
#   tools/experiments/fast_tile_variants.sh "TH TW QCAP" "TH TW QCAP" ...      (TH <= 30: the queue entries hold the row in 5 bits; TW a multiple of 16)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/visual_odometry_amd/csrc
for v in "$@"; do
  set -- $v; th=$1; tw=$2; qc=$3
  D=/tmp/ftv_${th}_${tw}_${qc}; mkdir -p $D
  ok=1
  for f in vo_api orb_kernels cv2order_kernels; do
    hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DFAST_TH=$th -DFAST_TW=$tw -DFT_QCAP=$qc -c $f.hip -o $D/$f.o 2>$D/$f.err || { echo "build failed: $f TH=$th TW=$tw QCAP=$qc"; tail -3 $D/$f.err; ok=0; }
  done
  [ $ok = 1 ] || continue
  hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libvo.so $D/vo_api.o $D/orb_kernels.o $D/cv2order_kernels.o match_kernels.o geom_kernels.o pnp_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o -ldl || continue
  echo "== FAST_TH $th  FAST_TW $tw  FT_QCAP $qc"
  ( cd $R; VO_HIP_LIBRARY=$D/libvo.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain --no-extras --no-faithful-pass $BENCH_ARGS | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  ',d['value'],d['ms_per_step'],d['stages']['fast_score_nms']['ms_per_launch'],d['stages']['select_fast']['ms_per_launch'],d['config'].get('pairs_ok_last_step'),d['config'].get('mean_inliers_last_step'))" )
done
