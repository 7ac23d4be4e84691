#!/bin/bash
# Diagnostic (GPU box): kernel timeline of the bench loop (default contexts) -> busy / idle / overlap report
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --steps 30 --warmup 4 "$@" > $O/log.txt 2>&1
cd $R
python3 $R/tools/experiments/trace_overlap.py $(find $O/t -name "*kernel_trace.csv" | head -1) 0.6 > $O/overlap.txt
cp $(find $O/t -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
rm -rf $O/t
cat $O/overlap.txt
