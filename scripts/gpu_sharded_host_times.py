"""Where does the host spend its time per tick on the sharded path (world 1, RCCL)?"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from cerebro_amd import capi
from cerebro_amd.sharded import ShardedLoopDetector
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
D = 4096
chip = capi.Chip(D, capacity_hint=rows + 2000)
chip.append_synthetic(rows + 1000, 1, [])
det = ShardedLoopDetector(chip, topk=8, device=torch.device("cuda", 0))
params = capi.default_dot_params()
T = {"scan_local": 0.0, "all_gather": 0.0, "merge_enq": 0.0, "collect": 0.0}
def run(n, l0):
    pend = []
    for i in range(n):
        l = l0 + 3 * i
        if len(pend) == 16:
            t = time.perf_counter(); det.collect(pend.pop(0)); T["collect"] += time.perf_counter() - t
        s = i % 16
        t = time.perf_counter(); st = chip.scan_local(l, det.local.data_ptr(), 8, params); T["scan_local"] += time.perf_counter() - t
        t = time.perf_counter()
        with torch.cuda.stream(det.stream):
            dist.all_gather_into_tensor(det.gathered, det.local)
        T["all_gather"] += time.perf_counter() - t
        t = time.perf_counter(); chip.merge_decide_enqueue(l, det.gathered.data_ptr(), 1, s, 8, params); T["merge_enq"] += time.perf_counter() - t
        pend.append(s)
    while pend:
        det.collect(pend.pop(0))
chip.loop_reset(); run(20, rows - 200); torch.cuda.synchronize(); chip.synchronize()
for k in T: T[k] = 0.0
chip.loop_reset()
t0 = time.perf_counter(); n = 200; run(n, rows - 800); torch.cuda.synchronize(); chip.synchronize(); dt = time.perf_counter() - t0
print(f"rows={rows}: {dt/n*1e6:.1f} us/tick;", {k: round(v / n * 1e6, 1) for k, v in T.items()})
det.close(); chip.close(); dist.destroy_process_group()
