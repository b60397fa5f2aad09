mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
grep -E "passed|failed|pytest_exit" gpurun_out/pytest_gpu.log
for i in 1 2 3; do
timeout 300 python bench.py --rows 10000 --steps 300 --warmup 20 --cpu-budget 0 --no-pnp --no-batch 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('10k:', round(j['value'],1),'ticks/s', round(j['ms_per_step']*1e3,1),'us/step')"
done
