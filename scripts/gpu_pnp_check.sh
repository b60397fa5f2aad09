mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pnp_gpu.py tests/test_fuzz_gpu.py tests/test_golden_frozen.py tests/test_config3_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -20
timeout 300 python scripts/gpu_pnp_fuzz.py 2>&1 | tail -1
bash scripts/gpu_pnp_ab.sh cerebro_amd/lib/ab/libcerebro_hip_v8.so cerebro_amd/lib/libcerebro_hip.so 3
bash scripts/gpu_pnp_stages.sh 50 | grep "stop=[230] "
