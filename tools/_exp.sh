#!/bin/bash
# usage: tools/_exp.sh "<extra hipcc defines>" tag  -- rebuild the library on the GPU box with extra defines and bench it
set -e
cd $GRAFT_REPO_ROOT
touch visual_odometry_amd/csrc/*.hip
make -C visual_odometry_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1" > gpurun_out/exp_build_$2.log 2>&1
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/exp_$2.json 2> gpurun_out/exp_$2.err
echo "== $2 ($1)"; python3 tools/bench_summary.py gpurun_out/exp_$2.json | head -2
