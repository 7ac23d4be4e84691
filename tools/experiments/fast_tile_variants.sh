#!/bin/bash
# Diagnostic (GPU box): k_fast with other tile heights / queue capacities (LDS per wave decides the waves per SIMD).  Every
# object is rebuilt with the same macros (the host geometry uses FAST_TH); each variant gets its own library copy.
#   tools/fast_tile_variants.sh "TH QCAP" "TH QCAP" ...      (TH <= 30: the queue entries hold the row in 5 bits)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R/visual_odometry_amd/csrc
for v in "$@"; do
  set -- $v; th=$1; qc=$2
  D=/tmp/ftv_${th}_${qc}; mkdir -p $D
  for f in vo_api orb_kernels match_kernels geom_kernels pnp_kernels cv2order_kernels gather_rccl jpeg_kernels sift_batch jpeg_host; do
    hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DFAST_TH=$th -DFT_QCAP=$qc -c $f.hip -o $D/$f.o 2>/dev/null || { echo "build failed: $f TH=$th QCAP=$qc"; continue 2; } &
  done; wait
  hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libvo.so $D/*.o -ldl || continue
  echo "== FAST_TH $th  FT_QCAP $qc"
  ( cd $R; VO_HIP_LIBRARY=$D/libvo.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain $BENCH_ARGS | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  ',d['value'],d['ms_per_step'],d['stages']['fast_score_nms']['ms_per_launch'],d['stages']['select_fast']['ms_per_launch'],d['config'].get('pairs_ok_last_step'),d['config'].get('mean_inliers_last_step'))" )
done
