cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(for i in 1 2; do
 for shape in "29000 4096" "29000 8192"; do
  for cfg in "CHIP_SCAN_DEPTH=1" "CHIP_SCAN_DEPTH=2" \
             "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=1" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" \
             "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=0" "CHIP_SCAN_ROWS=2" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_ROWS=2" \
             "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" "CHIP_SCAN_BLOCK=256 CHIP_SCAN_BPC=4 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=1"; do
    echo -n "[$shape $cfg] "; env CHIP_SCAN_PLAIN_MIB=2048 $cfg python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_shape_sweep.txt
for cfg in "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=1"; do
  echo "== $cfg"; python scripts/gpu_scan_stamps_xcd.py 29000 $cfg 2>&1 | grep -v Warning | tail -14
done | tee gpurun_out/r06/scan_stamps_29k_1024.txt
