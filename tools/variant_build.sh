#!/bin/bash
# Diagnostic (GPU box): rebuild ONE source with extra macros into a PRIVATE library copy (the product library is never touched;
# the copy is selected with VO_HIP_LIBRARY) and run a command with it — by default the bench, printing one stage.
#   tools/variant_build.sh geom_kernels "-DRS_WAVES=8" ["-DRS_WAVES=6" ...]
#   env: BENCH_ARGS (extra bench.py flags), STAGE (stage to print, default essential_ransac), RUN (a command to run instead of the bench)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/visual_odometry_amd/csrc
src=$1; shift
st=${STAGE:-essential_ransac}
i=0
for fl in "$@"; do
  i=$((i+1))
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $fl -c $src.hip -o /tmp/var_$src.o || exit 1
  objs=""; for f in vo_api orb_kernels match_kernels geom_kernels pnp_kernels cv2order_kernels gather_rccl jpeg_kernels sift_batch jpeg_host; do
    if [ $f = $src ]; then objs="$objs /tmp/var_$src.o"; else objs="$objs $f.o"; fi; done
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_var$i.so $objs -ldl || exit 1
  echo "== $src $fl"
  if [ -n "$RUN" ]; then ( cd $R; VO_HIP_LIBRARY=/tmp/libvo_var$i.so $RUN ); continue; fi
  ( cd $R; VO_HIP_LIBRARY=/tmp/libvo_var$i.so python3 bench.py --no-cpu-baseline --no-stream-pass --no-sustain --no-faithful-pass --no-extras $BENCH_ARGS | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('  ',d['value'],d['ms_per_step'],'$st',round(d['stages']['$st']['ms_total']/5,4))" )
done
