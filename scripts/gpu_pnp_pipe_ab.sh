# same-box A/B of the pipelined PnP launch form (CHIP_PNP_PIPE_SLOTS = slots per group; 0 = one launch pair): alternating runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2; do
  for cfg in "CHIP_PNP_PIPE_SLOTS=0" "CHIP_PNP_PIPE_SLOTS=512" "CHIP_PNP_PIPE_SLOTS=1024" "CHIP_PNP_PIPE_SLOTS=2048" "CHIP_PNP_PIPE_SLOTS=1024 CHIP_PNP_PIPE_FLAT=1" "CHIP_PNP_PIPE_SLOTS=256" "CHIP_PNP_GROUPS=2"; do
    echo -n "[$cfg] "; env $cfg timeout 300 python scripts/gpu_pnp_rates.py 2>&1 | tail -1
  done
done | tee gpurun_out/r06/pnp_pipe_ab.txt
