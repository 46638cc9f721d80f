#!/bin/bash
# what 4-byte beta columns could save at most (GPU box): a diagnostic build stores / reads the columns as floats (values rounded: results off), kernel times only
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
touch htk_amd/csrc/fb_lr.hip htk_amd/csrc/fb_kernels.hip
HTKAMD_LR_DEFS="-DLR_EXP_BUILD=1" python3 -m htk_amd.build > /dev/null 2>&1 || echo build failed
python3 tools/lr_exp.py "0 4096 1 0 4096" 2>&1 | tail -12
touch htk_amd/csrc/fb_lr.hip htk_amd/csrc/fb_kernels.hip; python3 -m htk_amd.build > /dev/null 2>&1
