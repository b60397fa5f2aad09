cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash scripts/gpu_pnp_pipe_ab.sh
(for shape in "29000 4096" "45000 4096" "60000 4096" "100000 4096" "15000 8192" "29000 8192"; do
  for cfg in "" "CHIP_SCAN_ROWS=1" "CHIP_SCAN_ROWS=2" "CHIP_SCAN_ROWS=3"; do
    echo -n "[$shape $cfg] "; env $cfg python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
done) | tee gpurun_out/r06/scan_rows_policy_ab.txt
(timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/r06/pytest_gpu.log); grep -E "passed|failed|pytest_exit|^FAILED|^ERROR" gpurun_out/r06/pytest_gpu.log | tail -15
