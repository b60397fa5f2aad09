mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
grep -E "passed|failed|pytest_exit" gpurun_out/pytest_gpu.log
bash scripts/gpu_db_sizes.sh
