#!/bin/bash
# Diagnostic (GPU box): what one pass of the Huffman decoder costs — private library builds (VO_HIP_LIBRARY; the product library
# is untouched) whose propagation loop is cut to N rounds (-DJPG_DBG_ROUNDS=N: results are WRONG, only the time matters).
#   all rounds / 1 round / 0 rounds  ->  kernels ms per 257-file batch; the differences are the cost of a counting pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cat > /tmp/_jpegk.sh <<'X'
VO_JPEG_NOCHECK=1 python3 tests/scripts/bench_jpeg.py --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   kernels ms per batch', d['gpu_decode_kernels_ms_per_batch'])"
X
RUN="bash /tmp/_jpegk.sh" $R/tools/variant_build.sh jpeg_kernels "-DJPG_DBG_ROUNDS=512" "-DJPG_DBG_ROUNDS=1" "-DJPG_DBG_ROUNDS=0" 2>&1 | grep -A1 "^=="
