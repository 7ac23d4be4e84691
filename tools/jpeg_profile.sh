#!/bin/bash
# Run on the GPU box: JPEG tests, the JPEG bench line and the rocprofv3 kernel summary of the same bench -> gpurun_out/jpeg/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/jpeg
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_jpeg.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tests/scripts/bench_jpeg.py "$@" > $O/jpeg_bench.json 2> $O/jpeg_bench.err || { tail -5 $O/jpeg_bench.err; exit 1; }
cat $O/jpeg_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tests/scripts/bench_jpeg.py --repeats 2 "$@" > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/jpeg_kernel_stats.csv && rm -rf $O/prof
grep -E "jpeg|Name|fillBuffer" $O/jpeg_kernel_stats.csv | cut -c1-160
