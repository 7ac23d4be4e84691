#!/bin/bash
# Diagnostic (GPU box): variants of jpeg_kernels.hip (private library builds, tools/variant_build.sh) — kernels' time for 257 and
# 1024 files per launch and the files-in -> poses-out pipeline (three contexts).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cat > /tmp/_jv.sh <<X
FRAMES=257 bash $R/tools/experiments/jpeg_kernels_ms.sh
FRAMES=1024 bash $R/tools/experiments/jpeg_kernels_ms.sh
python3 $R/tests/scripts/bench_jpeg_pipeline.py | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   pipeline pairs/s', d['frame_pairs_per_s'])"
X
V=("-DJPG_NCK=2" "-DJPG_NCK=1 -DJPG_CK0_SHIFT=2" "-DJPG_NCK=0")
[ $# -gt 0 ] && V=("$@")
RUN="bash /tmp/_jv.sh" $R/tools/variant_build.sh jpeg_kernels "${V[@]}" 2>&1 | grep -A3 "^=="
