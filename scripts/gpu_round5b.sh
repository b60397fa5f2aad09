#!/bin/bash
# Final-tree evidence of round 5 after the resident scan instance: GPU suite, default bench, the resident tests and timelines.
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD:$PWD/tests
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python -m pytest tests/test_resident_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee $OUT/resident_pytest.log
L=cerebro_amd/lib/sync_tick_latency
{
for rep in 1 2; do
  for rows in 5000 10000 20000 29000; do
    echo "launched rows=$rows: $(timeout 120 $L $rows 3000)"
    echo "resident rows=$rows: $(CHIP_TICK_RESIDENT=1 timeout 120 $L $rows 3000)"
  done
done
for rows in 10000 29000; do
  echo "relay, line in VRAM (CHIP_RESIDENT_BAR=1) rows=$rows: $(CHIP_TICK_RESIDENT=1 CHIP_RESIDENT_BAR=1 timeout 120 $L $rows 3000)"
  echo "relay, line pinned  (CHIP_RESIDENT_BAR=0) rows=$rows: $(CHIP_TICK_RESIDENT=1 CHIP_RESIDENT_BAR=0 timeout 120 $L $rows 3000)"
  echo "launched 10 Hz rows=$rows: $(timeout 200 $L $rows 100 0 100)"
  echo "resident 10 Hz rows=$rows: $(CHIP_TICK_RESIDENT=1 timeout 200 $L $rows 100 0 100)"
done
for r in 10000 29000; do timeout 300 python scripts/gpu_resident_stamps.py $r 2>&1 | grep -v amdgpu.ids; done
for r in 10000; do CHIP_RESIDENT_BAR=1 timeout 300 python scripts/gpu_resident_stamps.py $r 2>&1 | grep -v amdgpu.ids | sed "s/^rows=/relay variant (CHIP_RESIDENT_BAR=1): rows=/"; done
} > $OUT/resident_latency.txt 2>&1
cat $OUT/resident_latency.txt | cut -c1-200
if [ "${1:-}" = "full" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_full.log 2>&1; grep -E "passed|failed|Error|^FAILED" $OUT/pytest_gpu_full.log | tail -8 | tee $OUT/pytest_gpu.log
  timeout 600 python bench.py > $OUT/bench_default.log 2>&1; tail -c 3000 $OUT/bench_default.log
fi
