#!/usr/bin/env python3
"""Host-side cost of a tick through the multi-GPU entry points, measured on ONE device (the only hardware the builder has):
  (a) chip_create_multi with G sub-contexts on device 0 and a SHORT prefix (GPU time negligible): time per
      chip_loop_tick_enqueue call and ticks/s of the pipelined loop -- what one host thread + G-1 worker threads must sustain
      (the 8-GPU target is one tick per ~0.31 ms: 125k rows/GPU);
  (b) the same DB size per device as the 8-way shard of the 1M DB (125k rows on the one device) through the plain ctx, the
      sharded ctx with the in-library RCCL exchange at world 1 (scan reserve 0 and 4), and a one-device group over RCCL."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from cerebro_amd import capi  # noqa: E402

D = 4096


def loop(chip, ls, W=16):
    p = capi.default_dot_params()
    pend, enq = [], 0.0
    t0 = time.perf_counter()
    for i, l in enumerate(ls):
        if len(pend) == W:
            chip.loop_tick_collect(pend.pop(0))
        a = time.perf_counter()
        chip.loop_tick_enqueue(l, i % W, p)
        enq += time.perf_counter() - a
        pend.append(i % W)
    while pend:
        chip.loop_tick_collect(pend.pop(0))
    dt = time.perf_counter() - t0
    return len(ls) / dt, 1e6 * enq / len(ls)


out = {"host_cost_short_prefix": {}, "rows_125k": {}}
for G in (1, 2, 4, 8):
    with capi.Chip(D, capacity_hint=8000, devices=[0] * G, copy_exchange=True) as chip:
        chip.append_synthetic(8000, 1)
        ls = [4000 + 3 * i for i in range(1300)]   # query rows stay inside the replicated ring (newest 4096 rows)
        loop(chip, ls[:200])
        chip.loop_reset()
        tps, enq_us = loop(chip, ls)
        out["host_cost_short_prefix"][f"G={G}"] = {"ticks_per_s": tps, "enqueue_us_per_tick": enq_us, "prefix_rows": 3950}

rows = 125_000
ls = [rows + 50 + 3 * i for i in range(600)]


def run(label, make, after=None):
    chip = make()
    if after:
        after(chip)
    chip.append_synthetic(ls[-1], 1)
    loop(chip, ls[:100])
    chip.loop_reset()
    tps, enq_us = loop(chip, ls)
    out["rows_125k"][label] = {"ticks_per_s": tps, "enqueue_us_per_tick": enq_us}
    chip.close()


run("plain ctx", lambda: capi.Chip(D, capacity_hint=ls[-1]))
for res in ("0", "4"):
    os.environ["CHIP_SCAN_RESERVE"] = res
    run(f"sharded ctx + in-library RCCL world 1, reserve {res}", lambda: capi.Chip(D, capacity_hint=ls[-1]),
        lambda c: c.comm_init_rank(capi.comm_unique_id(), 1, 0))
del os.environ["CHIP_SCAN_RESERVE"]
run("group of one device over RCCL (ncclCommInitAll)", lambda: capi.Chip(D, capacity_hint=ls[-1], devices=[0]))
run("group of 8 sub-contexts on one device (8 x 15.6k rows each, copy exchange)", lambda: capi.Chip(D, capacity_hint=ls[-1], devices=[0] * 8))
sys.stdout.flush()
print(json.dumps(out, indent=1), flush=True)
os.dup2(2, 1)   # whatever C stdio still holds (the RCCL banner) leaves through stderr, not after the JSON
