#!/bin/bash
# variants of gmm_exact.hip built ON the GPU box, each run through tools/dec_diag.py: bash tools/r05_exvar.sh "<defs>" ...
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for defs in "$@"; do
   touch htk_amd/csrc/gmm_exact.hip
   HTKAMD_EX_DEFS="$defs" python3 -m htk_amd.build > gpurun_out/exvar_build.log 2>&1 || { echo "build failed: $defs"; tail -5 gpurun_out/exvar_build.log; continue; }
   echo "== $defs"
   timeout 600 python3 tools/dec_diag.py 256 2>&1 | tail -1
done
