#!/bin/bash
# PMC passes for the matcher kernels (run on the GPU box).  $1 = mfma | mfma_fp4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pmc_match; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --steps 2 --warmup 1 --matcher-kernel ${1:-mfma}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/b -- $B > $O/b.log 2>&1
cd $R
python3 $R/tools/pmc_summary.py $O/a k_nn > $O/k_nn_counters.txt
python3 $R/tools/pmc_summary.py $O/b k_nn >> $O/k_nn_counters.txt
rm -rf $O/a $O/b
cat $O/k_nn_counters.txt; tail -3 $O/b.log
