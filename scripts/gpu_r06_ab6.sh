# early re-issue at raised priority (CHIP_SCAN_DEPTH=3: depth 1, =4: depth 2) vs the product stream (=1) and depth 2 (=2); alternating runs, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(for i in 1 2 3; do
 for shape in "29000 4096" "20000 4096" "45000 4096" "15000 8192"; do
  for d in 1 3 2 4; do
    echo -n "[$shape CHIP_SCAN_DEPTH=$d] "; env CHIP_SCAN_DEPTH=$d python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_prio_ab.txt
for d in 3 4; do echo "== CHIP_SCAN_DEPTH=$d"; python scripts/gpu_scan_stamps_xcd.py 29000 CHIP_SCAN_DEPTH=$d 2>&1 | grep -v Warning | tail -14; done | tee gpurun_out/r06/scan_stamps_29k_prio.txt
CHIP_SCAN_DEPTH=3 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_golden_8d.py tests/test_f64_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06/pytest_depth3.log 2>&1; tail -3 gpurun_out/r06/pytest_depth3.log
CHIP_SCAN_DEPTH=4 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_golden_8d.py tests/test_f64_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06/pytest_depth4.log 2>&1; tail -3 gpurun_out/r06/pytest_depth4.log
