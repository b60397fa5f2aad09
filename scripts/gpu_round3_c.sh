# round 3, call C: group robustness (owner-only bulk append, failure mark, owner row fetch) + half-occupancy launches for cache-sized prefixes
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_c.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_c.log); tail -25 gpurun_out/pytest_c.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 600 python scripts/gpu_short_scan.py --set half --rows 10000 --ticks 900 > gpurun_out/half_scan.txt 2> gpurun_out/half_scan.err; cat gpurun_out/half_scan.txt
