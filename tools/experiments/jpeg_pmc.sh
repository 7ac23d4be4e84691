#!/bin/bash
# Run on the GPU box: SQ counters of the JPEG kernels -> gpurun_out/jpeg/jpeg_pmc_sq.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/jpeg
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- python3 $R/tests/scripts/bench_jpeg.py --repeats 1 "$@" > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $O/sq2 -- python3 $R/tests/scripts/bench_jpeg.py --repeats 1 "$@" > $O/sq2.log 2>&1
cd $R
python3 $R/tools/pmc_summary.py $O/sq k_jpeg > $O/jpeg_pmc_sq.txt 2>&1; python3 $R/tools/pmc_summary.py $O/sq2 k_jpeg >> $O/jpeg_pmc_sq.txt 2>&1
rm -rf $O/sq $O/sq2
cat $O/jpeg_pmc_sq.txt | head -90
