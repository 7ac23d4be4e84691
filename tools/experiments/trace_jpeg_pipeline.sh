#!/bin/bash
# Diagnostic (GPU box): kernel timeline of the files-in -> poses-out pipeline (tests/scripts/bench_jpeg_pipeline.py) -> busy / overlap report
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/trace_jpeg; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tests/scripts/bench_jpeg_pipeline.py "$@" > $O/log.txt 2>&1
cd $R
python3 $R/tools/experiments/trace_overlap.py $(find $O/t -name "*kernel_trace.csv" | head -1) 0.5 > $O/overlap.txt
rm -rf $O/t
tail -2 $O/log.txt | cut -c1-400
cat $O/overlap.txt
