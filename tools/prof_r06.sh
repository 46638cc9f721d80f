#!/bin/bash
# Round-6 profile set, written under gpurun_out/<tag>/ (then: python tools/prof_summarise.py gpurun_out/<tag> <tag>):
#   1-6. as tools/prof_r05.sh: kernel trace of the default bench command, FETCH_SIZE / WRITE_SIZE / SQ / MFMA / TCC passes, the plain bench line
#   7.   the decode leg's kernels (tools/dec_diag.py) and the front end's (tools/mfcc_bench.py) under the same counters: what bench.py's
#        other_paths rooflines cite (k_decode's traffic; k_mfcc_frames' instruction / LDS figures)
# usage (GPU box): bash tools/prof_r06.sh r06b
set -uo pipefail
tag=${1:-r06}
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
bash tools/r06_pmc_cmd.sh "$tag/dec" k_decode tools/dec_diag.py 256 > "$out/dec_pmc.log" 2>&1 || echo "decode counters failed"
bash tools/r06_pmc_cmd.sh "$tag/mfcc" k_mfcc_frames tools/mfcc_bench.py > "$out/mfcc_pmc.log" 2>&1 || echo "mfcc counters failed"
cd "${GRAFT_REPO_ROOT}"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 600 "$out/bench.json"
