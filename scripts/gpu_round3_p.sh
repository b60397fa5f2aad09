#!/bin/bash
# round 3, call P: factor-wave issue priority A/B (CHIP_PNP_PRIO=0/1): stamps, 1000-hyp call time, batch-8 rate, bit-exactness
mkdir -p gpurun_out
for pr in 0 1; do
  echo "=== CHIP_PNP_PRIO=$pr"
  CHIP_PNP_PRIO=$pr timeout 300 python scripts/gpu_pnp_stamps.py 50 2>&1 | tail -14
  for i in 1 2; do CHIP_PNP_PRIO=$pr timeout 300 python scripts/gpu_pnp_rates.py 2>&1 | tail -1; done
done
timeout 900 python -m pytest tests/test_pnp_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python scripts/gpu_pnp_fuzz.py 2>&1 | tail -3
