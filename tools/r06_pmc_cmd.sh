#!/bin/bash
# counters of one kernel of ANY python command (GPU box): bash tools/r06_pmc_cmd.sh <tag> <kernel-name-prefix> <script.py> [args...]
# separate passes (MI355X_MICROARCH.md "rocprofv3 PMC slots"): FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ LDS / waits; per-launch means -> <tag>/summary.json
set -uo pipefail
tag=$1; kn=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/$tag; mkdir -p "$out"
rocprofv3 --kernel-trace --stats -d "$out/t" -o p --output-format csv -- python3 "$@" > "$out/t.out" 2> "$out/t.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/f" -o p --output-format csv -- python3 "$@" > /dev/null 2> "$out/f.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/w" -o p --output-format csv -- python3 "$@" > /dev/null 2> "$out/w.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$out/s" -o p --output-format csv -- python3 "$@" > /dev/null 2> "$out/s.log"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d "$out/s2" -o p --output-format csv -- python3 "$@" > /dev/null 2> "$out/s2.log"
python3 - "$out" "$kn" <<'P'
import csv, glob, sys, collections, json
out, kn = sys.argv[1:3]
res = {"kernel_prefix": kn}
def match(name): return name.startswith(kn) or name.split(" ", 1)[-1].startswith(kn)
fs = glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True)
if fs:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(fs[0])) if match(r["Kernel_Name"])]
    if d: res["duration_us"] = {"n": len(d), "mean": sum(d) / len(d), "median": sorted(d)[len(d) // 2], "min": min(d)}
for dname in ("f", "w", "s", "s2"):
    fs = glob.glob(f"{out}/{dname}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if match(r["Kernel_Name"]): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): res[k] = {"n": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res))
P
