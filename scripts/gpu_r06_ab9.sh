# rows claimed by PAIRS of workgroups (older + younger workgroup of a CU) at depth 2 (tuning build, CHIP_SCAN_DEPTH=7): parity, stamps, rates
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=$GRAFT_REPO_ROOT/cerebro_amd/lib/tune/libcerebro_hip.so
CHIP_SCAN_DEPTH=7 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_golden_8d.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06/pytest_depth7.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r06/pytest_depth7.log | tail -3
for cfg in "CHIP_SCAN_DEPTH=7" "CHIP_SCAN_DEPTH=1"; do
  echo "== $cfg"; python scripts/gpu_scan_stamps_xcd.py 29000 $cfg 2>&1 | grep -v "Warning\|nanmean\|amdgpu.ids" | tail -13
done | tee gpurun_out/r06/scan_stamps_29k_pair.txt
(for i in 1 2 3; do
 for shape in "29000 4096" "20000 4096" "45000 4096" "15000 8192"; do
  for d in 1 7; do
    echo -n "[$shape CHIP_SCAN_DEPTH=$d] "; env CHIP_SCAN_DEPTH=$d python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_pair_ab.txt
