#!/bin/bash
# round-6 checkpoint 2 (GPU box): shim + MFCC tests, MFCC counters, decode counters, shim latency
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_htklib_shim.py tests/test_gpu_parity.py tests/test_wave.py tests/test_quals.py -x -q -m gpu -k "shim or mfcc or wav or qual" 2>&1 | tail -5
python tools/mfcc_bench.py 2>&1 | tail -2
python tools/shim_latency.py 100 2>&1 | tail -1
bash tools/r06_pmc_cmd.sh r06_mfcc_pmc k_mfcc_frames tools/mfcc_bench.py 2>&1 | tail -1
