mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_scan_gpu.py tests/test_concurrency_gpu.py tests/test_host_replay.py tests/test_f64_gpu.py -m gpu -q -x > gpurun_out/pytest_j.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_j.log); tail -6 gpurun_out/pytest_j.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
cat > /tmp/fused_ab.py <<'PY'
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
import gpu_short_scan as g
ref = None
for rows in (10000, 30000):
    for env in ({"CHIP_TICK_FUSED": 0}, {"CHIP_TICK_FUSED": 1}, {"CHIP_TICK_FUSED": 1, "CHIP_SCAN_SHORT_BPC": 0}, {"CHIP_TICK_FUSED": 1, "CHIP_SCAN_STREAMS": 2}):
        os.environ.pop("CHIP_TICK_FUSED", None)
        r, sig = g.run_config(rows, env, 900, 16)
        print(json.dumps(r), flush=True)
PY
timeout 600 python /tmp/fused_ab.py > gpurun_out/fused_ab.txt 2> gpurun_out/fused_ab.err; cat gpurun_out/fused_ab.txt | cut -c1-260; tail -3 gpurun_out/fused_ab.err
