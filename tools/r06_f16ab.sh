cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_phases" 2>&1 | tail -3
for v in "" "HTKAMD_NO_TAPER_SKIP=1"; do
  env $v python bench.py --score fastest --cpu-seconds 0 --extras 0 --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['kernel_ms'])"
done
