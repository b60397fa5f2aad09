mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_l.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_l.log); tail -6 gpurun_out/pytest_l.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
