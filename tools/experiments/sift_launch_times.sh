#!/bin/bash
# Diagnostic (GPU box): duration of every SIFT kernel launch of one single-context step, in launch order (which octaves / tap counts
# cost what) -> gpurun_out/sift_launches.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/sift_trace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --detector sift --contexts 1 --steps 2 --warmup 1 --no-cpu-baseline --no-stream-pass --no-sustain --no-faithful-pass --no-extras --no-profile > $O/log.txt 2>&1
cd $R
python3 - "$O" <<'PY' > gpurun_out/sift_launches.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sb = [r for r in rows if 'k_sb_' in r['Kernel_Name']]
n = len(sb) // 3                                    # three steps (1 warm-up + 2)
last = sb[-n:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:>5}x{r['Grid_Size_Z']:>3}  {r['Kernel_Name'][:60]}")
PY
rm -rf $O
