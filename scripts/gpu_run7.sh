mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -12 gpurun_out/pytest_gpu.log
