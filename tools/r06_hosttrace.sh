#!/bin/bash
# where the host's time of an EM iteration goes on THIS box (bench.py's BENCH_HOST_TRACE): launch | wait for the update | collect | update begin | next tables
cd "${GRAFT_REPO_ROOT:?}"
grep -m1 "model name" /proc/cpuinfo; python3 -c "import os; print('cpus', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
BENCH_HOST_TRACE=1 HTKAMD_PREP_TIMING=1 python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 40 2> gpurun_out/hosttrace.err > gpurun_out/hosttrace.json
grep "host it" gpurun_out/hosttrace.err | tail -30 | awk '{a+=$4; b+=$5; c+=$6; d+=$7; e+=$8; n++} END {printf "mean over %d iterations: launch %.3f  wait-update %.3f  collect %.3f  update-begin %.3f  next-tables %.3f ms\n", n, a/n, b/n, c/n, d/n, e/n}'
grep "prepare workers" gpurun_out/hosttrace.err | tail -20 | awk '{a+=$3; n++} END {printf "prepare workers %.3f ms\n", a/n}'
grep "prepare stage" gpurun_out/hosttrace.err | tail -20 | awk '{a+=$3; n++} END {printf "prepare stage+copy %.3f ms\n", a/n}'
python3 -c "import json; d=json.loads(open('gpurun_out/hosttrace.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['kernel_ms'])"
for n in 0 8 0 8; do HTKAMD_EXTRA_FILLS=$n python3 bench.py --cpu-seconds 0 --extras 0 --also-fastest 0 --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extra fills $n: ms_per_step', round(d['ms_per_step'],4))"; done
