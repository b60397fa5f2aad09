mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_pnp_gpu.py tests/test_config3_gpu.py tests/test_fuzz_gpu.py tests/test_golden_frozen.py -m gpu -q -x > gpurun_out/pytest_n.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_n.log); tail -5 gpurun_out/pytest_n.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 600 python scripts/gpu_pnp_fuzz.py > gpurun_out/pnp_fuzz_n.txt 2>&1; tail -4 gpurun_out/pnp_fuzz_n.txt
timeout 300 python scripts/gpu_pnp_stage.py > gpurun_out/pnp_stage_n.txt 2>&1; grep "stop=0\|stop=1" gpurun_out/pnp_stage_n.txt
timeout 300 python scripts/gpu_pnp_batch_perf.py > gpurun_out/pnp_batch_n.txt 2>&1; tail -5 gpurun_out/pnp_batch_n.txt
timeout 300 python scripts/gpu_pnp_stamps.py 1000 > gpurun_out/pnp_stamps_n.txt 2>&1; tail -7 gpurun_out/pnp_stamps_n.txt
