mkdir -p gpurun_out
(timeout 500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -15 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench1.log 2>&1; echo exit=$? >> gpurun_out/bench1.log)
tail -5 gpurun_out/bench1.log
for v in "512 2 0" "512 3 0" "256 4 0" "256 6 0" "512 2 1" "512 2 2" "256 8 1"; do set -- $v; echo "== block=$1 bpc=$2 variant=$3"; CHIP_SCAN_BLOCK=$1 CHIP_SCAN_BPC=$2 CHIP_SCAN_VARIANT=$3 timeout 300 python bench.py --steps 30 --warmup 3 --cpu-budget 0 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    try: j=json.loads(line)
    except Exception: print(line.rstrip()); continue
    print(round(j['value'],1),'ticks/s', round(j['roofline']['achieved'],1),'GB/s', round(j['roofline']['avg_kernel_ms'],4),'ms')
"; done > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
