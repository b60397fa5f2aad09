"""Pipelined tick rate of db_scan_topk at 1M / 500k / 125k / 100k / 10k rows for the CURRENT environment (CHIP_SCAN_* knobs,
CHIP_LIB=<path> to time another build of the library) -- run ONE process per configuration: contexts created later in a process
share hardware queues with the earlier ones and time slower.  Prints a checksum of a few query results so that A/B variants
can be seen to agree.  Used for profiles/r02_scan_load_path.txt."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa
from cerebro_amd import capi
if os.environ.get("CHIP_LIB"):
    capi.LIB_PATH = Path(os.environ["CHIP_LIB"]).resolve()
import bench

D = 4096
params = capi.default_dot_params()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CHIP_SCAN"))
chip = capi.Chip(D, capacity_hint=1_001_500)
chip.append_synthetic(1_001_500, bench.SEED, [(1_000_200, 4321, 1), (1_000_201, 99_000, 1)])
got = [chip.query_rows(k, [1_000_200, 1_000_201, 1_000_202], 8) for k in (1_000_000, 100_003, 777, 17)]
chk = hash(tuple(np.concatenate([np.concatenate([a[0].view(np.int64).ravel(), a[1].ravel()]) for a in got]).tolist()))
sizes = os.environ.get("SIZES")
for rows, n in ([(int(x), max(60, min(600, 120_000_000 // int(x)))) for x in sizes.split(",")] if sizes else ((1_000_000, 110), (500_000, 200), (125_000, 300), (100_000, 300), (10_000, 600))):
    ls = [rows + bench.LAG + 3 * i for i in range(n)]
    chip.loop_reset()
    bench.run_ticks(chip, ls[:20], params, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_ticks(chip, ls[20:], params, 16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n - 20)
    print(f"[{tag}] chk {chk & 0xffffff:06x} rows {rows:8d}: {dt*1e6:8.1f} us/tick  {4.0*D*rows/dt/1e12:6.3f} TB/s", flush=True)
chip.close()
