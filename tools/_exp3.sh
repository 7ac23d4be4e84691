#!/bin/bash
# usage: tools/_exp3.sh "<defines>" tag -- rebuild everything on the box, print FAST stage time + value
cd $GRAFT_REPO_ROOT
touch visual_odometry_amd/csrc/*.hip
make -C visual_odometry_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1" > gpurun_out/exp_build_$2.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stream-pass --steps 6 --warmup 2 > gpurun_out/exp_$2.json 2> gpurun_out/exp_$2.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/exp_$2.json')); print('$2', 'fast', d['stages']['fast_score_nms']['ms_per_launch'], 'sel', d['stages']['select_fast']['ms_per_launch'], 'value', d['value'], 'inl', d['config']['mean_inliers_last_step'])"
