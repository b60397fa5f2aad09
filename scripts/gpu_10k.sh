for i in 1 2 3; do
timeout 300 python bench.py --rows 10000 --steps 300 --warmup 20 --cpu-budget 0 --no-pnp --no-batch 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('10k:', round(j['value'],1),'ticks/s', round(j['ms_per_step']*1e3,1),'us/step', round(j['roofline']['avg_kernel_ms']*1e3,1))"
done
