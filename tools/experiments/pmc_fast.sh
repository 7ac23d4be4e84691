#!/bin/bash
# PMC passes for k_fast only (run on the GPU box): issue / LDS / wait counters of the dominant kernel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pmc_fast; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d $O/b -- $B > $O/b.log 2>&1
cd $R
python3 $R/tools/pmc_summary.py $O/a k_fast > $O/k_fast_counters.txt
python3 $R/tools/pmc_summary.py $O/b k_fast >> $O/k_fast_counters.txt
rm -rf $O/a $O/b
cat $O/k_fast_counters.txt
