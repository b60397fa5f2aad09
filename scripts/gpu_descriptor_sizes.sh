# descriptor size of the reference's default model (8192) at the same DB bytes as the 1M x 4096 headline config
(timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -3 gpurun_out/pytest_gpu.log
for cfg in "4096 1000000" "8192 500000" "10240 400000" "2048 2000000"; do set -- $cfg
  timeout 300 python bench.py --dim $1 --rows $2 --steps 60 --warmup 5 --cpu-budget 0 --no-pnp --no-batch 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('D=$1 rows=$2:', round(j['value'],1),'ticks/s', round(j['roofline']['achieved'],0),'GB/s kernel', round(j['roofline']['avg_kernel_ms'],3),'ms')"
done
