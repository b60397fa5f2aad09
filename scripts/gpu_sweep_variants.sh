# scan-kernel A/B variants (library built with -DCHIP_SCAN_TUNING_VARIANTS): 0 production, 1-6 unroll / rows in flight / plain loads,
# 7-10 other cache-policy bits on the streaming loads (sc1 nt, sc0 sc1 nt, sc0 nt, sc1)
mkdir -p gpurun_out
for v in ${VARIANTS:-0 7 8 9 10 0}; do
  for rep in 1 2; do
  CHIP_SCAN_VARIANT=$v timeout 300 python bench.py --steps 60 --warmup 5 --cpu-budget 0 --no-pnp --no-batch 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    try: j=json.loads(line)
    except Exception: continue
    print('variant $v', round(j['value'],1),'ticks/s', round(j['roofline']['achieved'],1),'GB/s', round(j['roofline']['avg_kernel_ms'],4),'ms')
"; done; done 2>&1 | tee gpurun_out/sweep.log
CHIP_SCAN_VARIANT=8 timeout 600 python -m pytest tests/test_scan_gpu.py -q -k "shapes or golden" 2>&1 | grep -E "passed|failed"
