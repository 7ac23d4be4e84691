#!/bin/bash
# usage: tools/_rep.sh tag n  -- run the bench n times with the current build, print ransac stage time and value
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $2); do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/rep_$1_$i.json 2> gpurun_out/rep_$1_$i.err || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/rep_$1_$i.json')); print('$1 $i ransac', d['stages']['essential_ransac']['ms_per_launch'], 'pose', d['stages']['recover_pose']['ms_per_launch'], 'value', d['value'])"
done
