#!/bin/bash
# usage: tools/_exp2.sh "<defines>" tag -- rebuild on the box, print the FAST stage time only
cd $GRAFT_REPO_ROOT
touch visual_odometry_amd/csrc/orb_kernels.hip
make -C visual_odometry_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1" > gpurun_out/exp_build_$2.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stream-pass --steps 3 --warmup 1 > gpurun_out/exp_$2.json 2> gpurun_out/exp_$2.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/exp_$2.json')); print('$2', 'fast', d['stages']['fast_score_nms']['ms_per_launch'], 'value', d['value'])"
