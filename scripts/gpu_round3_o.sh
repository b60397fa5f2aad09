mkdir -p gpurun_out
for g in 1 2; do echo "CHIP_PNP_GROUPS=$g"; CHIP_PNP_GROUPS=$g timeout 300 python scripts/gpu_pnp_batch_perf.py 2>&1 | tail -5; done > gpurun_out/pnp_groups.txt; cat gpurun_out/pnp_groups.txt
CHIP_PNP_GROUPS=2 timeout 300 python -m pytest tests/test_pnp_gpu.py tests/test_config3_gpu.py -m gpu -q 2>&1 | tail -2
