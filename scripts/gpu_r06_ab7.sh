# one 512-thread workgroup per CU (no second workgroup to starve) with deeper load streams: stamps + rates (tuning build of the library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=$GRAFT_REPO_ROOT/cerebro_amd/lib/tune/libcerebro_hip.so
for cfg in "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=1" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=5"; do
  echo "== $cfg"; python scripts/gpu_scan_stamps_xcd.py 29000 $cfg 2>&1 | grep -v Warning | tail -14
done | tee gpurun_out/r06/scan_stamps_29k_half.txt
(for i in 1 2; do
 for shape in "29000 4096" "10000 4096"; do
  for cfg in "CHIP_SCAN_DEPTH=1" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=5"; do
    echo -n "[$shape $cfg] "; env $cfg python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_half_ab.txt
