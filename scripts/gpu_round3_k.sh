mkdir -p gpurun_out
cat > /tmp/mid_ab.py <<'PY'
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
import gpu_short_scan as g
for rows in (29000, 60000):
    for env in ({"CHIP_SCAN_ROWS": -1}, {"CHIP_SCAN_ROWS": 1}, {"CHIP_SCAN_ROWS": 2}, {"CHIP_SCAN_ROWS": 3}, {"CHIP_SCAN_PLAIN_MIB": 2048},
                {"CHIP_SCAN_PLAIN_MIB": 2048, "CHIP_SCAN_SHORT_BPC": 0}, {"CHIP_SCAN_ROWS": -1, "CHIP_TICK_SAME_STREAM": 0, "CHIP_SCAN_STREAMS": 2}):
        r, sig = g.run_config(rows, env, 600, 16)
        print(json.dumps(r), flush=True)
PY
timeout 900 python /tmp/mid_ab.py > gpurun_out/mid_ab.txt 2> gpurun_out/mid_ab.err; cut -c1-230 gpurun_out/mid_ab.txt; tail -2 gpurun_out/mid_ab.err
