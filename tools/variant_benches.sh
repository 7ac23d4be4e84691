#!/bin/bash
# Run on the GPU box: the bench lines of the non-default configurations -> gpurun_out/variants/*.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/variants; rm -rf $O; mkdir -p $O
cd $R
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-sustain --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; return; }
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:22s} {d['value']:9.1f} pairs/s  step {d['ms_per_step']:.3f} ms  ransac {d['stages']['essential_ransac']['ms_per_launch']:.3f} ms  iters mean {d['config']['ransac_iters']['mean']}")
PY
}
run canonical_order --keypoint-order canonical
run pair_stride6 --pair-stride 6
run hard_pairs --distinct-frames 17 --matcher crosscheck-legacy
run opencv300 --poly-solver opencv300
run ratio_matcher --matcher ratio
run config3_1080p --width 1920 --height 1080 --nfeatures 4000 --nlevels 4 --pairs-per-step 128
run kitti_shape --width 1241 --height 376
run independent_pairs --workload independent
run matcher_int8 --matcher-kernel mfma
run matcher_popcount --matcher-kernel popcount
run sift_1152x648 --detector sift --width 1152 --height 648
run sift_independent --detector sift --workload independent --pairs-per-step 96
