#!/bin/bash
# round-5: the device update with and without the fused element + gConst kernel (GPU box)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/${1:-r05upd}
mkdir -p "$out"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu > "$out/pytest.log" 2>&1
echo "pytest rc=$?" | tee "$out/rc.txt"; tail -5 "$out/pytest.log"
for u in 1 0 1 0 1 0; do
   if [ $u = 1 ]; then export HTKAMD_UPD_UNFUSED=1; else unset HTKAMD_UPD_UNFUSED; fi
   timeout 600 python bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 30 > "$out/bench_unfused$u.json" 2> "$out/bench_unfused$u.err"
   echo "bench unfused=$u rc=$?" | tee -a "$out/rc.txt"
   python - "$out/bench_unfused$u.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "parts", json.dumps(d.get("em_iteration_parts_ms")), "oracle", json.dumps(d.get("oracle_check"))[:400])
except Exception as e:
    print("no bench line:", e)
P
done
