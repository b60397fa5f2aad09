(timeout 900 python -m pytest tests/test_pnp_gpu.py tests/test_icp_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3)
python scripts/gpu_pnp_stage.py 2>&1 | grep "H=" | grep -E "stop=(2|3|0)"
