mkdir -p gpurun_out
timeout 300 python scripts/gpu_pnp_stamps.py 1000 > gpurun_out/pnp_stamps.txt 2>&1; cat gpurun_out/pnp_stamps.txt | grep -v amdgpu.ids
timeout 300 python scripts/gpu_pnp_stamps.py 50 >> gpurun_out/pnp_stamps.txt 2>&1; tail -9 gpurun_out/pnp_stamps.txt | grep -v amdgpu.ids
