"""Isolate what serialises ticks on the sharded path: (a) ctx-owned stream, no exchange; (b) torch stream, no exchange;
(c) torch stream + D2D copy as the exchange; (d) torch stream + RCCL all_gather (world 1)."""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
from cerebro_amd import capi
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
chip = capi.Chip(4096, capacity_hint=rows + 2000)
chip.append_synthetic(rows + 1000, 1, [])
params = capi.default_dot_params()
local = torch.zeros((3, 8, 2), dtype=torch.float64, device="cuda")
gathered = torch.zeros((3, 8, 2), dtype=torch.float64, device="cuda")
stream = torch.cuda.Stream()
def run(n, l0, mode):
    pend = []
    for i in range(n):
        l = l0 + 3 * i
        if len(pend) == 16: chip.loop_tick_collect(pend.pop(0))
        s = i % 16
        chip.scan_local(l, local.data_ptr(), 8, params)
        src = local
        if mode == "copy":
            with torch.cuda.stream(stream): gathered.copy_(local, non_blocking=True)
            src = gathered
        elif mode == "nccl":
            with torch.cuda.stream(stream): dist.all_gather_into_tensor(gathered, local)
            src = gathered
        chip.merge_decide_enqueue(l, src.data_ptr(), 1, s, 8, params)
        pend.append(s)
    while pend: chip.loop_tick_collect(pend.pop(0))
for name, use_torch_stream, mode in [("a ctx stream, no exchange", False, "none"), ("b torch stream, no exchange", True, "none"),
                                     ("c torch stream + copy", True, "copy"), ("d torch stream + all_gather", True, "nccl")]:
    chip.set_stream(stream.cuda_stream if use_torch_stream else None)
    chip.loop_reset(); run(20, rows - 200, mode); torch.cuda.synchronize(); chip.synchronize()
    chip.loop_reset(); t0 = time.perf_counter(); n = 300; run(n, rows - 950, mode); torch.cuda.synchronize(); chip.synchronize()
    print(f"rows={rows} {name}: {(time.perf_counter()-t0)/n*1e6:.1f} us/tick")
chip.set_stream(None); chip.close(); dist.destroy_process_group()
