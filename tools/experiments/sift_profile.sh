#!/bin/bash
# Run on the GPU box: SIFT tests, bench line and the rocprofv3 kernel summary of the same bench -> gpurun_out/sift/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/sift
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_sift.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log | cut -c1-250; exit 1; }
tail -1 $O/tests.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tests/scripts/bench_sift.py "$@" > $O/sift_bench.json 2> $O/sift_bench.err || { tail -5 $O/sift_bench.err; exit 1; }
cat $O/sift_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tests/scripts/bench_sift.py --oracle-frames 0 "$@" > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/sift_kernel_stats.csv && rm -rf $O/prof
grep -E "sift|Name" $O/sift_kernel_stats.csv | cut -c1-150
