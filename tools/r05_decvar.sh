#!/bin/bash
# variants of decode.hip built ON the GPU box, each run through tools/dec_diag.py: bash tools/r05_decvar.sh "<defs>" "<defs>" ...
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for defs in "$@"; do
   touch htk_amd/csrc/decode.hip
   HTKAMD_DEC_DEFS="$defs" python3 -m htk_amd.build > gpurun_out/decvar_build.log 2>&1 || { echo "build failed: $defs"; tail -5 gpurun_out/decvar_build.log; continue; }
   echo "== $defs"
   timeout 600 python3 tools/dec_diag.py 256 2>&1 | tail -2
done
