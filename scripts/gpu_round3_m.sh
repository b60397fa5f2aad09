mkdir -p gpurun_out
timeout 300 python scripts/gpu_pnp_stamps.py 1000 > gpurun_out/pnp_stamps_m.txt 2>&1; tail -9 gpurun_out/pnp_stamps_m.txt
timeout 300 python scripts/gpu_pnp_stamps.py 50 > gpurun_out/pnp_stamps_m50.txt 2>&1; tail -8 gpurun_out/pnp_stamps_m50.txt
