#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/${1:-r05m}; mkdir -p "$out"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$out/pytest.log"
timeout 600 python tests/fuzz_parity.py 150 20261103 > "$out/fuzz.log" 2>&1; echo "fuzz rc=$?"; tail -3 "$out/fuzz.log"
for v in 1 0; do
   if [ $v = 1 ]; then export HTKAMD_NO_MIXSTATE=1; else unset HTKAMD_NO_MIXSTATE; fi
   timeout 600 python bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 30 > "$out/bench_nomix$v.json" 2> "$out/bench_nomix$v.err"
   python - "$out/bench_nomix$v.json" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "kernel_ms", d.get("kernel_ms")); print(json.dumps(d.get("oracle_check",{}).get("accumulators_max_rel_diff"))[:500])
P
done
