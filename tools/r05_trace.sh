#!/bin/bash
# kernel trace of a short bench run, with the update's kernels listed in launch order for one iteration (GPU box)
set -uo pipefail
tag=${1:-r05t}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/$tag; mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o "$tag" --output-format csv -- python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 20 > "$out/bench.json" 2> "$out/rocprof.log" || echo "trace failed"
python3 - "$out" "$tag" <<'P'
import csv, glob, sys, collections
out, tag = sys.argv[1:3]
f = glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last EM iteration: from the last k_upd_mark back to the previous one
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_upd_mark")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
with open(f"{out}/iteration_timeline.txt", "w") as o:
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        o.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  {r['Kernel_Name'][:90]}\n")
print(open(f"{out}/iteration_timeline.txt").read())
by = collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10: print(f"{sum(v)/len(v):9.1f} us avg  {sorted(v)[len(v)//2]:9.1f} med  n={len(v):4d}  {k[:80]}")
P
