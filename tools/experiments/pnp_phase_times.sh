#!/bin/bash
# Diagnostic (GPU box): where k_pnp_ransac spends its time — private library builds (VO_HIP_LIBRARY; the product library is
# untouched) cut short with -DVO_PNP_STOP=n (1: after the 64 EPnP hypotheses of the first round, 2: after scoring + the final
# inlier mask, 3: after the DLT / homography start of cv2's final solvePnP, 0: full kernel incl. the LM refinement).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cat > /tmp/_pnpk.sh <<'X'
python3 tests/scripts/bench_pnp.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   kernel ms per launch', d['kernel_ms_per_launch'])"
X
RUN="bash /tmp/_pnpk.sh" $R/tools/variant_build.sh pnp_kernels "-DVO_PNP_STOP=1" "-DVO_PNP_STOP=2" "-DVO_PNP_STOP=3" "-DVO_PNP_STOP=0" 2>&1 | grep -A1 "^=="
