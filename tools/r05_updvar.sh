#!/bin/bash
# variants of update.hip built ON the GPU box and traced: bash tools/r05_updvar.sh "<defs A>" "<defs B>" ...
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
n=0
for defs in "$@"; do
   n=$((n+1)); out=gpurun_out/updvar$n; mkdir -p "$out"
   touch htk_amd/csrc/update.hip
   HTKAMD_UPD_DEFS="$defs" python3 -m htk_amd.build > "$out/build.log" 2>&1 || { echo "build failed: $defs"; tail -5 "$out/build.log"; continue; }
   rocprofv3 --kernel-trace --stats -d "$out/trace" -o v --output-format csv -- python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 30 > "$out/bench.json" 2> "$out/rocprof.log" || echo "trace failed"
   echo "== $defs"
   python3 - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)): by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    if "upd" in k or "build_bf16" in k: print(f"{sum(v)/len(v):9.1f} us avg {sorted(v)[len(v)//2]:9.1f} med {min(v):9.1f} min n={len(v):4d}  {k[:70]}")
P
done
