mkdir -p gpurun_out/prof5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo smoke_exit=$? >> gpurun_out/smoke.log); tail -3 gpurun_out/smoke.log
(timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_pnp.log 2>&1; echo exit=$? >> gpurun_out/bench_pnp.log)
python - <<'PY'
import json
for line in open('gpurun_out/bench_pnp.log'):
    if line.startswith('{'):
        j=json.loads(line); print(json.dumps(j['pnp'],indent=1)); print('ticks/s',j['value'], 'cpu', j.get('cpu_baseline'))
    elif 'exit=' in line or 'Error' in line or 'error' in line: print(line.rstrip())
PY
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5/trace -o r01c -- python bench.py --rows 10000 --steps 5 --warmup 1 --cpu-budget 0 > gpurun_out/prof5/trace.log 2>&1
