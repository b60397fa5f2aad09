mkdir -p gpurun_out
for rows in 10000 100000 125000 1000000; do
  timeout 300 python bench.py --rows $rows --steps 200 --warmup 20 --cpu-budget 0 --no-pnp 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('single ', $rows, round(j['value'],1),'ticks/s', round(j['ms_per_step']*1e3,1),'us/step  kernel', round(j['roofline']['avg_kernel_ms']*1e3,1),'us', round(j['roofline']['achieved'],0),'GB/s')"
  timeout 300 python bench.py --rows $rows --steps 200 --warmup 20 --cpu-budget 0 --no-pnp --force-sharded 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('sharded', $rows, round(j['value'],1),'ticks/s', round(j['ms_per_step']*1e3,1),'us/step  kernel', round(j['roofline']['avg_kernel_ms']*1e3,1),'us')"
done
