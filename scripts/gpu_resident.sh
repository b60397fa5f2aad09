#!/bin/bash
# GPU session of the resident scan instance (round 5): its tests, the scan suite on the refactored row-batched kernel, and the
# synchronous-tick latency from C -- launched (this build / the build before the refactor) vs resident.  Output: gpurun_out/r05/resident_*.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05; mkdir -p $OUT
export PYTHONPATH=$PWD:$PWD/tests
timeout 900 python -m pytest tests/test_resident_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/resident_pytest.log
cat $OUT/resident_pytest.log
if [ "${1:-}" != "quick" ]; then
  timeout 900 python -m pytest tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/resident_scan_suite.log
  cat $OUT/resident_scan_suite.log
fi
L=cerebro_amd/lib/sync_tick_latency
{
for rep in 1 2; do
  echo "launched  rows=10000: $(timeout 120 $L 10000 3000)"
  for rows in 5000 10000; do
    echo "resident  rows=$rows: $(CHIP_TICK_RESIDENT=1 timeout 120 $L $rows 3000)"
  done
  echo "res+claim rows=10000: $(CHIP_TICK_RESIDENT=1 CHIP_SCAN_CLAIM=1 timeout 120 $L 10000 3000)"
done
for r in 10000 5000; do timeout 300 python scripts/gpu_resident_stamps.py $r; done
} > $OUT/resident_latency.txt 2>&1
cat $OUT/resident_latency.txt
