#!/bin/bash
# round-5 experiment driver (GPU box): parity of the forward-backward paths, then the bench line with the lean kernels on and off.
# usage: bash tools/r05_try.sh <tag> [quick]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r05x}
out=gpurun_out/$tag
mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$out/pytest_parity.log" 2>&1
echo "pytest_parity rc=$?" | tee "$out/rc.txt"
tail -5 "$out/pytest_parity.log"
if [ "${2:-}" != "quick" ]; then
   timeout 900 python tests/fuzz_parity.py 250 20261101 > "$out/fuzz.log" 2>&1
   echo "fuzz rc=$?" | tee -a "$out/rc.txt"
   tail -8 "$out/fuzz.log"
fi
for m in ${LEANS:-0 7 15}; do
   HTKAMD_LR_LEAN=$m timeout 600 python bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 30 > "$out/bench_lean$m.json" 2> "$out/bench_lean$m.err"
   echo "bench lean=$m rc=$?" | tee -a "$out/rc.txt"
   python - "$out/bench_lean$m.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "kernel_ms", d.get("kernel_ms"), "oracle", json.dumps(d.get("oracle_check"))[:600])
except Exception as e:
    print("no bench line:", e)
P
   tail -3 "$out/bench_lean$m.err"
done
