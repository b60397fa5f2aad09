cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(for i in 1 2 3; do
 for shape in "10000 4096" "5000 8192" "20000 4096"; do
  for cfg in "CHIP_SCAN_DEPTH=1" "CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=1" "CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2" "CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_SHORT_BPC=0" "CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_HALF_MIB=400"; do
    echo -n "[$shape $cfg] "; env $cfg python scripts/gpu_shape_ab.py $shape 2>&1 | tail -1
  done
 done
done) | tee gpurun_out/r06/scan_depth_small.txt
