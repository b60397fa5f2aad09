# does a start-up skew between the waves of a workgroup lift the one-workgroup-per-CU shapes off 6.6-6.9 TB/s?  (tuning build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=$GRAFT_REPO_ROOT/cerebro_amd/lib/tune/libcerebro_hip.so
for cfg in "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_STAGGER=5" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=5" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=10" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=20"; do
  echo "== $cfg"; python scripts/gpu_scan_stamps_xcd.py 29000 $cfg 2>&1 | grep -v "Warning\|nanmean\|amdgpu.ids" | tail -13
done | tee gpurun_out/r06/scan_stamps_29k_stagger.txt
(for i in 1 2; do
  for cfg in "CHIP_SCAN_DEPTH=1" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=5" "CHIP_SCAN_BLOCK=1024 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=20" "CHIP_SCAN_BLOCK=512 CHIP_SCAN_BPC=1 CHIP_SCAN_CLAIM=1 CHIP_SCAN_DEPTH=2 CHIP_SCAN_STAGGER=10"; do
    echo -n "[29000 4096 $cfg] "; env $cfg python scripts/gpu_shape_ab.py 29000 4096 2>&1 | tail -1
  done
done) | tee gpurun_out/r06/scan_stagger_ab.txt
