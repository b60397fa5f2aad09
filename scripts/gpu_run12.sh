(timeout 900 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -3)
python scripts/gpu_batch_perf.py 2>&1 | grep "rows=1000000"
