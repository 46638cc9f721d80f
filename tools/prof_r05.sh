#!/bin/bash
# Round-5 profile set of the default bench command, written under gpurun_out/<tag>/ (then: python tools/prof_summarise.py gpurun_out/<tag> <tag>):
#   1. rocprofv3 --kernel-trace --stats            per-kernel durations of the timed EM iterations (and of the other_paths legs)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE          HBM-side bytes per launch (separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   3. --pmc SQ counters                           instruction mix / wait share of the recursion and statistics kernels
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES ...          matrix-pipe utilisation of the scoring kernel
#   5. --pmc TCC_HIT_sum TCC_MISS_sum              where the scoring kernel's table tiles come from (L2 hit rate)
#   6. the plain bench line (with the CPU legs)
# usage (GPU box): bash tools/prof_r05.sh r05a
set -uo pipefail
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set: run through gpurun}"
out=gpurun_out/$tag
mkdir -p "$out"
one="--cpu-seconds 0 --extras 0 --also-fastest 0 --steps 1 --warmup 0"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o "$tag" --output-format csv -- python3 bench.py --cpu-seconds 0 > "$out/bench_under_rocprof.json" 2> "$out/rocprof_trace.log" || echo "trace pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
   rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_$c" -o "$tag" --output-format csv -- python3 bench.py $one > "$out/bench_pmc_$c.json" 2> "$out/rocprof_$c.log" || echo "$c pass failed"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$out/pmc_SQ" -o "$tag" --output-format csv -- python3 bench.py $one > "$out/bench_pmc_SQ.json" 2> "$out/rocprof_SQ.log" || echo "SQ pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA --kernel-trace -d "$out/pmc_MFMA" -o "$tag" --output-format csv -- python3 bench.py $one > "$out/bench_pmc_MFMA.json" 2> "$out/rocprof_MFMA.log" || echo "MFMA pass failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$out/pmc_TCC" -o "$tag" --output-format csv -- python3 bench.py $one > "$out/bench_pmc_TCC.json" 2> "$out/rocprof_TCC.log" || echo "TCC pass failed"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 800 "$out/bench.json"
