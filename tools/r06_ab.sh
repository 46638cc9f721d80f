#!/bin/bash
# short bench lines of runtime / build variants ON the GPU box: bash tools/r06_ab.sh "<name>|<ENV=value or ->" ...   (an ENV that htk_amd/build.py reads also rebuilds)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/ab
for spec in "$@"; do
   IFS='|' read -r name envs <<< "$spec"
   if [ "$envs" != "-" ]; then export "${envs%%=*}=${envs#*=}"; fi
   case "$envs" in HTKAMD_*_DEFS=*|HTKAMD_B16_*) touch htk_amd/csrc/*.hip; python3 -m htk_amd.build > "gpurun_out/ab/$name.build.log" 2>&1 || { echo "build failed: $spec"; continue; } ;; esac
   python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps ${STEPS:-60} > "gpurun_out/ab/$name.json" 2> "gpurun_out/ab/$name.err" || { echo "bench failed: $spec"; tail -3 "gpurun_out/ab/$name.err"; }
   python3 - "$name" "gpurun_out/ab/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-14s ms_per_step %.4f  kernel_ms %s  score_work %s" % (sys.argv[1], d["ms_per_step"], json.dumps(d.get("kernel_ms")), json.dumps(d.get("score_work"))))
except Exception as e:
    print(sys.argv[1], "no bench line:", e)
P
   if [ "$envs" != "-" ]; then unset "${envs%%=*}"; fi
   case "$envs" in HTKAMD_*_DEFS=*|HTKAMD_B16_*) touch htk_amd/csrc/*.hip; python3 -m htk_amd.build > /dev/null 2>&1 ;; esac
done
