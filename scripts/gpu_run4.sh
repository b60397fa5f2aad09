mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_pnp_gpu.py -x -q > gpurun_out/pytest_pnp.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_pnp.log)
tail -40 gpurun_out/pytest_pnp.log
