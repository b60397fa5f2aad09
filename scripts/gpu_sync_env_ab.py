import json, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from gpu_sync_tick import run
for rows in (10000, 29000):
    out, res = run(rows, {}, n=400)
    print(os.environ.get("HSA_ENABLE_INTERRUPT"), os.environ.get("GPU_MAX_HW_QUEUES"), json.dumps(out), flush=True)
