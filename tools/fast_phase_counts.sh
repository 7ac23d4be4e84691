#!/bin/bash
# Diagnostic (GPU box): dynamic VALU / LDS / SALU instruction counts of k_fast up to each phase boundary, from PMC runs
# of builds cut short with -DFT_STOP_AFTER=n (results are wrong in those builds; only the counters matter).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/fast_phases; rm -rf $O; mkdir -p $O
cd $R/visual_odometry_amd/csrc
cp ../libvo_hip.so /tmp/libvo_hip.keep
for n in 1 2 3 4 0; do
  if [ $n -eq 0 ]; then D=""; else D="-DFT_STOP_AFTER=$n"; fi
  hipcc -O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function $D -c orb_kernels.hip -o /tmp/orb_v.o || exit 1
  hipcc -shared -fPIC --offload-arch=gfx950 -o ../libvo_hip.so vo_api.o /tmp/orb_v.o match_kernels.o geom_kernels.o pnp_kernels.o cv2order_kernels.o gather_rccl.o jpeg_kernels.o sift_kernels.o -ldl
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/v$n -- python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --steps 2 --warmup 1 > $O/v$n.log 2>&1 )
  echo "== stop after phase $n (0 = full kernel)"; python3 $R/tools/pmc_summary.py $O/v$n k_fast | grep -E "INSTS|IDX"
  python3 - <<PY
import csv,glob
d=[float(r["End_Timestamp"])-float(r["Start_Timestamp"]) for p in glob.glob("$O/v$n/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(p)) if "k_fast" in r["Kernel_Name"]]
print("   k_fast mean duration us", sum(d)/len(d)/1e3 if d else None)
PY
  rm -rf $O/v$n
done
cp /tmp/libvo_hip.keep ../libvo_hip.so
