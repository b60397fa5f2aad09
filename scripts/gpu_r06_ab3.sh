# depth-2 claimed row stream (CHIP_SCAN_DEPTH=2) against depth 1, alternating runs on one box; rows form extended to 2 GiB prefixes for the A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(for i in 1 2; do
 for shape in "29000 4096" "20000 4096" "45000 4096" "60000 4096" "100000 4096" "29000 8192" "15000 8192"; do
  for cfg in "CHIP_SCAN_PLAIN_MIB=2048 CHIP_SCAN_DEPTH=1" "CHIP_SCAN_PLAIN_MIB=2048 CHIP_SCAN_DEPTH=2"; do
    echo -n "[$shape $cfg] "; env $cfg python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_depth_ab.txt
CHIP_SCAN_PLAIN_MIB=2048 CHIP_SCAN_DEPTH=2 timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_golden_8d.py tests/test_f64_gpu.py tests/test_resident_gpu.py -m gpu -q -x 2>&1 | tail -5
CHIP_SCAN_PLAIN_MIB=2048 CHIP_SCAN_DEPTH=2 CHIP_TICK_RESIDENT=1 timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_resident_gpu.py -m gpu -q -x 2>&1 | tail -5
python scripts/gpu_scan_stamps_xcd.py 29000 CHIP_SCAN_DEPTH=2 2>&1 | tail -25 | tee gpurun_out/r06/scan_stamps_29k_depth2.txt
python scripts/gpu_scan_stamps_xcd.py 29000 CHIP_SCAN_DEPTH=1 2>&1 | tail -25 | tee gpurun_out/r06/scan_stamps_29k_depth1.txt
