#!/bin/bash
# build variants ON the GPU box and trace the short bench: bash tools/r05_var.sh <kernel-name-filter-regex> "<ENV=defs>" "<ENV=defs>" ...
# ENV is one of HTKAMD_LR_DEFS (fb_lr.hip, fb_kernels.hip), HTKAMD_UPD_DEFS (update.hip), HTKAMD_B16_DEFS (gmm_bf16.hip); "" = the plain build
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
filt=$1; shift
n=0
for spec in "$@"; do
   n=$((n+1)); out=gpurun_out/var$n; mkdir -p "$out"
   touch htk_amd/csrc/update.hip htk_amd/csrc/fb_lr.hip htk_amd/csrc/fb_kernels.hip htk_amd/csrc/gmm_bf16.hip
   if [ -n "$spec" ]; then env "${spec%%=*}=${spec#*=}" python3 -m htk_amd.build > "$out/build.log" 2>&1; else python3 -m htk_amd.build > "$out/build.log" 2>&1; fi
   if [ $? -ne 0 ]; then echo "build failed: $spec"; tail -5 "$out/build.log"; continue; fi
   rocprofv3 --kernel-trace --stats -d "$out/trace" -o v --output-format csv -- python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 30 > "$out/bench.json" 2> "$out/rocprof.log" || echo "trace failed"
   echo "== $spec"
   python3 - "$out" "$filt" <<'P'
import csv, glob, sys, collections, re, json
out, filt = sys.argv[1:3]
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)): by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    if re.search(filt, k): print(f"{sum(v)/len(v):9.1f} us avg {sorted(v)[len(v)//2]:9.1f} med {min(v):9.1f} min n={len(v):4d}  {k[:70]}")
try:
    d = json.loads(open(f"{out}/bench.json").read().strip().splitlines()[-1]); print("ms_per_step", round(d["ms_per_step"], 4))
except Exception as e: print("no bench line", e)
P
done
