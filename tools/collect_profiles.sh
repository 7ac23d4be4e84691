#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 evidence bench.py and DESIGN.md cite into
# gpurun_out/prof_<tag><suffix>/; the summaries are then copied into profiles/ (committed).
#   tools/collect_profiles.sh <tag> [orb|sift]
#   rocprofv3 --kernel-trace --stats          -> per-kernel average durations (single context: no co-running stream)
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE   -> HBM traffic per launch and per step (separate passes, MI355X_MICROARCH.md)
#   rocprofv3 --pmc SQ_*                      -> VALU / SALU / LDS instruction counts, wave cycles, issue stalls
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
DET=${2:-orb}
if [ $DET = sift ]; then SUF=_sift; DARGS="--detector sift"; FR=192; else SUF=""; DARGS=""; FR=257; fi
O=$R/gpurun_out/prof_$TAG$SUF
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-faithful-pass --no-extras $DARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 8 --warmup 2 --no-profile > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 3 --warmup 1 --no-profile > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 3 --warmup 1 --no-profile > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B --steps 3 --warmup 1 --no-profile > $O/sq.log 2>&1
cd $R
# static opcode classes of the dominant kernel, from this build's own ISA (bench.py prices the issue bound with them)
if [ $DET = sift ]; then echo "{}" > $O/isa_mix.json; else
  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -S visual_odometry_amd/csrc/orb_kernels.hip -o $O/orb_kernels.s
  python3 -c "import json,subprocess,sys; m=json.loads(subprocess.check_output([sys.executable,'tools/isa_mix.py','$O/orb_kernels.s','_Z6k_fastILb0EE','--json'])); json.dump({'k_fast<false>': m}, open('$O/isa_mix.json','w'))"
  rm -f $O/orb_kernels.s
fi
python3 tools/collect_traffic.py $O/fetch $O/write $O/${TAG}_pmc_traffic$SUF.json $FR $O/sq 4 $O/isa_mix.json > /dev/null
python3 tools/pmc_summary.py $O/sq > $O/${TAG}_pmc_sq_counters$SUF.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_final_kernel_stats$SUF.csv
cp $O/${TAG}_pmc_traffic$SUF.json profiles/${TAG}_pmc_traffic$SUF.json      # bench.py reads it for roofline.traffic
python3 bench.py $DARGS > $O/${TAG}_final_bench$SUF.json 2> $O/bench.err
rm -rf $O/stats $O/fetch $O/write $O/sq
ls -la $O
