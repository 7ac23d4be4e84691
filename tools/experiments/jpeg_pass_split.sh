#!/bin/bash
# Diagnostic (GPU box): where k_jpeg_huffman's time goes — private library builds (tools/variant_build.sh): the product build,
# no checkpoints (every propagation round decodes whole subsequences), threads per file, the writing pass without its
# stores, no writing pass at all.  (The last two decode nothing useful: the comparison with libjpeg is switched off for them.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export VO_JPEG_NOCHECK=1
V=("-DJPG_NCK=2" "-DJPG_NCK=0" "-DJPG_NT=768" "-DJPG_NT=1024" "-DJPG_NT=384" "-DJPG_EXP_NOSTORE" "-DJPG_EXP_NOWRITEPASS")
[ $# -gt 0 ] && V=("$@")
RUN="bash $R/tools/experiments/jpeg_kernels_ms.sh" $R/tools/variant_build.sh jpeg_kernels "${V[@]}" 2>&1 | grep -A1 "^=="
