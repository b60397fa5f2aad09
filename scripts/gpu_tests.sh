mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -15 gpurun_out/pytest_gpu.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
