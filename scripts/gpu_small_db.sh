for rows in 4096 8192 16384 32768 65536; do
  timeout 300 python bench.py --rows $rows --steps 200 --warmup 20 --cpu-budget 0 --no-pnp 2>&1 | grep '^{' | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print($rows, 'rows:', round(j['ms_per_step']*1e3,1),'us/step  kernel', round(j['roofline']['avg_kernel_ms']*1e3,2),'us')"
done
