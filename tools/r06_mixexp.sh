#!/bin/bash
# ablations of k_mixstate (GPU box): LR_EXP bits 256 no sums, 1024 no rows, 2048 no distances, 8192 no atomics, and combinations
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
touch htk_amd/csrc/fb_lr.hip htk_amd/csrc/fb_kernels.hip
HTKAMD_LR_DEFS="-DLR_EXP_BUILD=1 ${1:-}" python3 -m htk_amd.build > /dev/null 2>&1 || echo build failed
python3 tools/lr_exp.py "0 8192 256 8448 1024 2048 11520" 2>&1 | grep "round 1"
touch htk_amd/csrc/fb_lr.hip htk_amd/csrc/fb_kernels.hip; python3 -m htk_amd.build > /dev/null 2>&1
