#!/bin/bash
# round-5 checkpoint run (GPU box): the whole -m gpu suite, a fuzz sweep, the default bench line.   usage: bash tools/r05_full.sh <tag> [fuzz iterations]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r05x}
out=gpurun_out/$tag
mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest_gpu rc=$?" | tee "$out/rc.txt"
tail -6 "$out/pytest_gpu.log"
timeout 1200 python tests/fuzz_parity.py ${2:-300} 20261102 > "$out/fuzz.log" 2>&1
echo "fuzz rc=$?" | tee -a "$out/rc.txt"
tail -4 "$out/fuzz.log"
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "bench rc=$?" | tee -a "$out/rc.txt"
python - "$out/bench.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "kernel_ms", d.get("kernel_ms"))
    print("oracle", json.dumps(d.get("oracle_check"))[:900])
    print("fastest", json.dumps(d.get("fastest_mode"))[:300])
except Exception as e:
    print("no bench line:", e)
P
tail -3 "$out/bench.err"
