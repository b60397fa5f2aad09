mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_pnp_gpu.py tests/test_scan_gpu.py tests/test_sharded_multiproc_gpu.py -m gpu -q > gpurun_out/pytest_f.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_f.log); tail -30 gpurun_out/pytest_f.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
