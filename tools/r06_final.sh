#!/bin/bash
# round-6 closing run (GPU box): what the decoder's block-wise demand would be, the new tests, a fuzz sweep of the two-phase pass, the profile set, the bench line with its counters in place
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r06c}
# (the decoder's block-wise demand: bash tools/r05_decvar.sh "-DDEC_NEED=1" "-DDEC_NEED=4" "-DDEC_NEED=16" -- measured once, docs/history.md)

python -m pytest tests/test_htklib_shim.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q -m gpu -k "shim or two_phases or sits_out or exchange_in_parts" 2>&1 | tail -3
python tests/fuzz_parity.py 400 20261104 fb,streams 2>&1 | tail -2
bash tools/prof_r06.sh "$tag" 2>&1 | tail -2
python3 tools/prof_summarise.py gpurun_out/$tag $tag > /dev/null 2>&1
cp profiles/${tag}_traffic.json profiles/r06_traffic.json; cp profiles/${tag}_legs.json profiles/r06_legs.json
python3 bench.py > gpurun_out/$tag/bench_final.json 2> gpurun_out/$tag/bench_final.err
tail -c 400 gpurun_out/$tag/bench_final.json
cp profiles/${tag}_*.json profiles/r06_traffic.json profiles/r06_legs.json profiles/${tag}_kernel_stats.csv gpurun_out/$tag/ 2>/dev/null
