#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "policy or replay" 2>&1 | tail -8
