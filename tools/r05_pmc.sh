#!/bin/bash
# counters of one kernel of the bench's iteration (GPU box): bash tools/r05_pmc.sh <tag> <kernel-name-prefix>
set -uo pipefail
tag=${1:-r05p}; kn=${2:-k_upd_gauss_fused}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/$tag; mkdir -p "$out"
one="--cpu-seconds 0 --extras 0 --also-fastest 0 --steps 2 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/f" -o p --output-format csv -- python3 bench.py $one > /dev/null 2> "$out/f.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/w" -o p --output-format csv -- python3 bench.py $one > /dev/null 2> "$out/w.log"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$out/s" -o p --output-format csv -- python3 bench.py $one > /dev/null 2> "$out/s.log"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --kernel-trace -d "$out/s2" -o p --output-format csv -- python3 bench.py $one > /dev/null 2> "$out/s2.log"
python3 - "$out" "$kn" <<'P'
import csv, glob, sys, collections
out, kn = sys.argv[1:3]
for d in ("f", "w", "s", "s2"):
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith(kn) or ("<" in r["Kernel_Name"] and r["Kernel_Name"].split(" ",1)[-1].startswith(kn)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(d, k, "n", len(v), "mean %.4g" % (sum(v)/len(v)))
P
