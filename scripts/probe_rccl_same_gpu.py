"""Probe: does the bundled RCCL accept two ranks on one device?  Kept as the record of why the multi-process GPU tests use
gloo as transport."""
import os
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(rank), device="cuda"); out = torch.zeros((8,), device="cuda")
    dist.all_gather_into_tensor(out, x); torch.cuda.synchronize()
    print("rank", rank, "OK", out.tolist(), flush=True)
except Exception as e:
    print("rank", rank, "FAILED:", str(e).splitlines()[0][:300], flush=True)
