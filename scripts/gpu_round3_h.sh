mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_pnp_gpu.py tests/test_config3_gpu.py -m gpu -q -x > gpurun_out/pytest_h.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_h.log); tail -4 gpurun_out/pytest_h.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 300 python scripts/gpu_pnp_stage.py > gpurun_out/pnp_stage_h.txt 2>&1; grep "stop=0\|stop=2\|stop=4" gpurun_out/pnp_stage_h.txt
timeout 300 python scripts/gpu_pnp_stamps.py 1000 > gpurun_out/pnp_stamps_h.txt 2>&1; grep -v amdgpu.ids gpurun_out/pnp_stamps_h.txt
timeout 300 python scripts/gpu_pnp_batch_perf.py > gpurun_out/pnp_batch_h.txt 2>&1; tail -5 gpurun_out/pnp_batch_h.txt
