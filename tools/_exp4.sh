#!/bin/bash
cd $GRAFT_REPO_ROOT
touch visual_odometry_amd/csrc/*.hip
make -C visual_odometry_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1" > gpurun_out/exp_build_$2.log 2>&1 || exit 1
timeout -k 10 200 python3 tools/_time_detect.py "$2"
