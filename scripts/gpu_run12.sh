#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "pnp or threeway or gate or icp" 2>&1 | tail -8
python scripts/gpu_pnp_batch_perf.py
