#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats summary of one bench configuration -> gpurun_out/<tag>_kernel_stats.csv
#   tools/kernel_stats.sh <tag> [bench args ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/_t_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-stream-pass --no-sustain --no-profile --no-faithful-pass --no-extras --steps 3 --warmup 1 "$@" > $O/${TAG}_trace.log 2>&1
find $O/_t_$TAG -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
rm -rf $O/_t_$TAG
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${TAG}_kernel_stats.csv")):
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "max_us", round(float(r["MaxNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
