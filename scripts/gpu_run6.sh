mkdir -p gpurun_out/prof6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -5 gpurun_out/pytest_gpu.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof6/trace -o r01d -- python bench.py --rows 10000 --steps 5 --warmup 1 --cpu-budget 0 > gpurun_out/prof6/trace.log 2>&1
grep '^{' gpurun_out/prof6/trace.log | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('pnp', j['pnp']['value'], 'hyp/s', j['pnp']['ms_per_call_1000_hyp'],'ms/1000', j['pnp']['reference_mode_ms_per_call'],'ms ref-mode')"
