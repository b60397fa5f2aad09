for cfg in "512 2" "512 3" "768 2" "1024 1" "1024 2" "256 4" "256 6"; do set -- $cfg
  for rep in 1 2; do
  CHIP_SCAN_BLOCK=$1 CHIP_SCAN_BPC=$2 timeout 300 python bench.py --steps 60 --warmup 5 --cpu-budget 0 --no-pnp 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    try: j=json.loads(line)
    except Exception: 
        if 'rror' in line: print(line.rstrip())
        continue
    print('block $1 bpc $2:', round(j['value'],1),'ticks/s', round(j['roofline']['achieved'],1),'GB/s', round(j['roofline']['avg_kernel_ms'],4),'ms')
"; done; done
