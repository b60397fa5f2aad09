mkdir -p gpurun_out/prof3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -8 gpurun_out/pytest_gpu.log
for rows in 10000 100000 1000000; do
  timeout 300 python bench.py --rows $rows --steps 200 --warmup 20 --cpu-budget 0 2>&1 | grep '^{' > gpurun_out/bench_rows_$rows.json
  python -c "
import json;j=json.load(open('gpurun_out/bench_rows_$rows.json'))
print($rows, round(j['value'],1),'ticks/s', round(j['ms_per_step']*1e3,1),'us/step', round(j['roofline']['achieved'],1),'GB/s kernel', round(j['roofline']['avg_kernel_ms']*1e3,1),'us kernel')"
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3/trace -o r01b -- python bench.py --rows 10000 --steps 100 --warmup 5 --cpu-budget 0 > gpurun_out/prof3/trace.log 2>&1
