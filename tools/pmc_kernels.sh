#!/bin/bash
# Diagnostic (GPU box): issue / LDS / wait counters and the kernel-trace summary of one bench configuration.
#   tools/pmc_kernels.sh <tag> <kernel name filter> [bench args ...]        -> gpurun_out/pmc_<tag>/{counters.txt,kernel_stats.csv}
# Counters are collected in their own runs (--kernel-trace only beside --pmc), the program itself follows `--`.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; FILT=$2; shift 2
O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --contexts 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --no-faithful-pass --no-extras --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $B > $O/t.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- $B > $O/b.log 2>&1
cd $R
find $O/t -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 tools/pmc_summary.py $O/a "$FILT" > $O/counters.txt
python3 tools/pmc_summary.py $O/b "$FILT" >> $O/counters.txt
rm -rf $O/a $O/b $O/t
head -40 $O/kernel_stats.csv | cut -c1-160
cat $O/counters.txt
