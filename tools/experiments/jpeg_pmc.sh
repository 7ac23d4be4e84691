#!/bin/bash
# Diagnostic (GPU box): issue / LDS / wait counters of the JPEG kernels on the bench batch (tests/scripts/bench_jpeg.py) -> gpurun_out/pmc_jpeg/counters.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/pmc_jpeg; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/tests/scripts/bench_jpeg.py --repeats 1 --frames ${FRAMES:-257}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- $B > $O/b.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/a jpeg > $O/counters.txt
python3 tools/pmc_summary.py $O/b jpeg >> $O/counters.txt
rm -rf $O/a $O/b
cat $O/counters.txt
