#!/bin/bash
# usage (on the GPU box, through gpurun): tools/variant_bench.sh "<extra -D defines>" <tag>
# Rebuilds libvo_hip.so with extra compile-time parameters (tile shapes, queue sizes, ...) and prints the per-stage
# event times of the detection chain (tools/time_detect.py).  How the tile sizes in vo_internal.h were chosen.
cd $GRAFT_REPO_ROOT
touch visual_odometry_amd/csrc/*.hip
make -C visual_odometry_amd/csrc CXXFLAGS="-O3 -fPIC -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1" > gpurun_out/exp_build_$2.log 2>&1 || exit 1
timeout -k 10 200 python3 tools/time_detect.py "$2"
