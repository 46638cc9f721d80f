#!/bin/bash
# MFCC checkpoint (GPU box): the front end's tests, the fuzz family, timings of the pair / one-frame kernels, counters
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_gpu_parity.py tests/test_wave.py tests/test_quals.py tests/test_gpu_decode.py -x -q -m gpu -k "mfcc or wav or qual" 2>&1 | tail -3
python tests/fuzz_parity.py 150 20261103 mfcc 2>&1 | tail -2
python tools/mfcc_diag.py 2>&1 | tail -1
python tools/mfcc_bench.py 2>&1 | tail -1
HTKAMD_MFCC_ONE_FRAME=1 python tools/mfcc_bench.py 2>&1 | tail -1
bash tools/r06_pmc_cmd.sh r06_mfcc_pmc3 k_mfcc_frames tools/mfcc_bench.py 2>&1 | tail -1 | cut -c1-1200
export HTKAMD_MFCC_ONE_FRAME=1
bash tools/r06_pmc_cmd.sh r06_mfcc_pmc3_one k_mfcc_frames tools/mfcc_bench.py 2>&1 | tail -1 | cut -c1-1200
