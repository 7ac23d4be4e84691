#!/bin/bash
# Diagnostic (GPU box): A/B of compile-time knobs under the three-context bench loop.  Each argument is "units:defines", e.g.
#   tools/experiments/define_variants.sh ":" "cv2order_kernels:-DCV_LDS_CAP=2048" "geom_kernels,vo_api:-DFP_LANES=32"
# The named units are rebuilt with the defines into a private library (/tmp), the others come from the tree's objects; the product
# library is never touched.  Two bench runs per variant (value, ms per step, sustained median, the stage times named in $STAGES).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
ALL="vo_api orb_kernels match_kernels geom_kernels pnp_kernels cv2order_kernels gather_rccl jpeg_kernels sift_batch"
export STAGES=${STAGES:-"cv2_keypoint_order essential_ransac fast_score_nms"}
cd $R/visual_odometry_amd/csrc
i=0
for v in "$@"; do
  i=$((i+1)); units=${v%%:*}; defs=${v#*:}
  D=/tmp/defvar_$i; mkdir -p $D; objs=""; ok=1
  for u in $ALL; do
    if [[ ",$units," == *",$u,"* ]]; then
      hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $defs -c $u.hip -o $D/$u.o 2>$D/$u.err || { echo "build failed: $u $defs"; tail -3 $D/$u.err; ok=0; }
      objs="$objs $D/$u.o"
    else objs="$objs $u.o"; fi
  done
  [ $ok = 1 ] || continue
  hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libvo.so $objs jpeg_host.o -ldl || continue
  echo "== [$units] $defs"
  for rep in 1 2; do
  ( cd $R; VO_HIP_LIBRARY=$D/libvo.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-extras --no-faithful-pass --sustain-repeats 1 --sustain-seconds 4 --steps 60 --warmup 6 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys,os;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['config'].get('sustained') or {}
print('  ',d['value'],d['ms_per_step'],'sustained',s.get('pairs_per_s_median'),' '.join('%s %.4f'%(k,d['stages'][k]['ms_per_launch']) for k in os.environ['STAGES'].split() if k in d['stages']),d['config'].get('mean_inliers_last_step'))" )
  done
done
