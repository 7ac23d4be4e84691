#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 evidence bench.py and DESIGN.md cite into
# gpurun_out/prof/; copy the summaries into profiles/ afterwards (tools/collect_profiles.sh prints the cp lines).
#   rocprofv3 --kernel-trace --stats          -> per-kernel average durations (single context: no co-running stream)
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE   -> HBM traffic per launch (separate passes, MI355X_MICROARCH.md)
#   rocprofv3 --pmc SQ_*                      -> VALU / SALU / LDS instruction counts, wave cycles, issue stalls
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 8 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 3 --warmup 1 --no-profile > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 3 --warmup 1 --no-profile > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B --steps 3 --warmup 1 --no-profile > $O/sq.log 2>&1
cd $R
python3 tools/collect_traffic.py $O/fetch $O/write $O/${TAG}_pmc_traffic.json 257 $O/sq > /dev/null
python3 tools/pmc_summary.py $O/sq > $O/${TAG}_pmc_sq_counters.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_final_kernel_stats.csv
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # bench.py reads it for roofline.traffic
python3 bench.py > $O/${TAG}_final_bench.json 2> $O/bench.err
rm -rf $O/stats $O/fetch $O/write $O/sq
ls -la $O
