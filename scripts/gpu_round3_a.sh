# round 3, call A: GPU suite with the row-batched scan kernel as the default for short scans, short/long scan A/B, PnP stage split + PMC
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -5 gpurun_out/pytest_gpu.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 600 python scripts/gpu_short_scan.py --rows 10000,100000 > gpurun_out/short_scan.txt 2> gpurun_out/short_scan.err
cat gpurun_out/short_scan.txt
timeout 600 python scripts/gpu_short_scan.py --set long --rows 1000000 --ticks 100 > gpurun_out/long_scan.txt 2> gpurun_out/long_scan.err
cat gpurun_out/long_scan.txt
timeout 300 python scripts/gpu_pnp_stage.py > gpurun_out/pnp_stage.txt 2>&1
cat gpurun_out/pnp_stage.txt
timeout 900 bash scripts/gpu_pnp_pmc.sh > gpurun_out/pnp_pmc.log 2>&1
tail -3 gpurun_out/pnp_pmc.log
