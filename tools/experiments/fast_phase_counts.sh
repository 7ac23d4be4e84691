#!/bin/bash
# Diagnostic (GPU box): dynamic VALU / LDS / SALU instruction counts of k_fast up to each phase boundary, from PMC runs of
# PRIVATE library builds cut short with -DFT_STOP_AFTER=n (results are wrong in those builds; only the counters matter).
# The product library is never touched: every build is linked to /tmp and selected with VO_HIP_LIBRARY.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/fast_phases; rm -rf $O; mkdir -p $O
cd $R/visual_odometry_amd/csrc
OBJS="vo_api.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_batch.o jpeg_host.o"
for n in 1 2 3 4 0; do
  if [ $n -eq 0 ]; then D=""; else D="-DFT_STOP_AFTER=$n"; fi
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $D -c orb_kernels.hip -o /tmp/orb_v$n.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libvo_phase$n.so /tmp/orb_v$n.o $OBJS -ldl || exit 1
  ( cd /tmp; export TMPDIR=/tmp VO_HIP_LIBRARY=/tmp/libvo_phase$n.so; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/v$n -- python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --no-faithful-pass --no-extras --steps 2 --warmup 1 > $O/v$n.log 2>&1 )
  echo "== stop after phase $n (0 = full kernel)"; python3 $R/tools/pmc_summary.py $O/v$n k_fast | grep -E "INSTS|IDX"
  rm -rf $O/v$n
done
