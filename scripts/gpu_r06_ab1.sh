cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(for i in 1 2; do
  for cfg in "" "CHIP_SCAN_PLAIN_MIB=1024" "CHIP_SCAN_PLAIN_MIB=1024 CHIP_SCAN_CLAIM=0" "CHIP_SCAN_ROWS=2" ; do
    echo -n "[$cfg] "; env $cfg python scripts/gpu_shape_ab.py 29000 8192 2>&1 | tail -1
  done
done) | tee gpurun_out/r06/shape_ab_8192.txt
bash scripts/gpu_round6.sh suite
