mkdir -p gpurun_out
for v in 0 1 2 3 4 5 6; do
  for rep in 1 2; do
  CHIP_SCAN_VARIANT=$v timeout 300 python bench.py --steps 60 --warmup 5 --cpu-budget 0 --no-pnp 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    try: j=json.loads(line)
    except Exception: continue
    print('variant $v', round(j['value'],1),'ticks/s', round(j['roofline']['achieved'],1),'GB/s', round(j['roofline']['avg_kernel_ms'],4),'ms')
"; done; done 2>&1 | tee gpurun_out/sweep.log
CHIP_SCAN_VARIANT=5 timeout 600 python -m pytest tests/test_scan_gpu.py -q -k "shapes or golden or 100k" 2>&1 | tail -2
