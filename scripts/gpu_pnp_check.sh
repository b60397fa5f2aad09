# PnP parity subset + fuzz + same-box A/B of the current build against saved builds: bash scripts/gpu_pnp_check.sh [<saved lib> ...]
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pnp_gpu.py tests/test_fuzz_gpu.py tests/test_golden_frozen.py tests/test_config3_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -20
timeout 300 python scripts/gpu_pnp_fuzz.py 2>&1 | tail -1
for L in "$@" cerebro_amd/lib/libcerebro_hip.so; do echo -n "$(basename $L): "; CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=$L CHIP_PNP_OCCUPANCY=1 python scripts/gpu_pnp_rates.py 2>&1 | grep "pnp occupancy"; done
bash scripts/gpu_pnp_ab.sh 3 "$@" cerebro_amd/lib/libcerebro_hip.so
bash scripts/gpu_pnp_stages.sh 50 | grep "stop=[230] "
