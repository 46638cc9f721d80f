#!/bin/bash
# parity figures of build variants ON the GPU box: bash tools/r06_parvar.sh "<name>|<ENV=defs or ->|<scoreMode>" ...
# ENV as in tools/r05_var.sh (HTKAMD_B16_DEFS, HTKAMD_EX_DEFS ...); a runtime variable (e.g. HTKAMD_BF16_CHUNKED=1) is simply exported too.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/parvar
for spec in "$@"; do
   IFS='|' read -r name envs mode <<< "$spec"
   touch htk_amd/csrc/gmm_bf16.hip htk_amd/csrc/gmm_exact.hip
   if [ "$envs" != "-" ]; then export "${envs%%=*}=${envs#*=}"; fi
   python3 -m htk_amd.build > "gpurun_out/parvar/$name.build.log" 2>&1 || { echo "build failed: $spec"; tail -5 "gpurun_out/parvar/$name.build.log"; continue; }
   timeout 900 python3 tools/headline_live.py "$name" "$mode" "gpurun_out/parvar/$name.json" 2> "gpurun_out/parvar/$name.err" | tail -2
   if [ "$envs" != "-" ]; then unset "${envs%%=*}"; fi
done
touch htk_amd/csrc/gmm_bf16.hip htk_amd/csrc/gmm_exact.hip
python3 -m htk_amd.build > /dev/null 2>&1
