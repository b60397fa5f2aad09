"""Synchronous ticks at the reference's 10 Hz cadence through the ctypes binding: launched vs resident instance, with and without a second
(large) ctx alive in the process -- why bench.py's paced figure can differ from examples/sync_tick_latency.cc's.
   python scripts/gpu_paced_ticks.py [rows=10000] [big_rows=0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cerebro_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
big_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = capi.default_dot_params()
p.min_new = -(1 << 30)
big = None
if big_rows:
    os.environ.pop("CHIP_TICK_RESIDENT", None)
    big = capi.Chip(4096, capacity_hint=big_rows)
    big.append_synthetic(big_rows, 1, [])
for mode in ("launched", "resident"):
    os.environ.pop("CHIP_TICK_RESIDENT", None)
    if mode == "resident":
        os.environ["CHIP_TICK_RESIDENT"] = "1"
    with capi.Chip(4096, capacity_hint=rows + 500) as c:
        c.append_synthetic(rows + 400, 777, [])
        ls = [rows + 50 + 3 * (i % 100) for i in range(200)]
        for l in ls[:100]:
            c.loop_tick(l, p)
        lat = []
        for l in ls[:60]:
            time.sleep(0.1)
            t1 = time.perf_counter()
            c.loop_tick(l, p)
            lat.append(time.perf_counter() - t1)
        lat = 1e6 * np.array(lat[3:])
        print(f"{mode:9s} rows={rows} big_ctx_rows={big_rows}: paced 10 Hz p50 {np.median(lat):6.1f} us  min {lat.min():6.1f}  p90 {np.percentile(lat, 90):6.1f}  max {lat.max():6.1f}   "
              f"sorted head/tail {np.sort(lat)[:3].round(1)} {np.sort(lat)[-3:].round(1)}")
if big is not None:
    big.close()
