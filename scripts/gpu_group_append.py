#!/usr/bin/env python3
"""Cold-start bulk append (Cerebro.cpp:133-161,1005: after loadStateFromDisk the first tick copies ALL columns) into a group of G
sub-contexts: every device is sent only the rows it owns + the newest CHIP_RING_ROWS of the batch, so the bytes over PCIe are
~1x the batch (round 2 sent the whole batch to every device: Gx).  Reports wall time and the host->device bytes implied."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from cerebro_amd import capi

D, N = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(1)
db32 = rng.standard_normal((N, D), dtype=np.float32)
db32 /= np.linalg.norm(db32, axis=1, keepdims=True)
for wire, arr in (("f64", db32.astype(np.float64)), ("f32", db32)):
    for G in (1, 2, 8):
        with capi.Chip(D, capacity_hint=N, devices=[0] * G) as chip:
            t0 = time.perf_counter()
            (chip.append_f64 if wire == "f64" else chip.append_f32)(arr)
            chip.synchronize()
            dt = time.perf_counter() - t0
            ring = min(N, 4096) * (G if G > 1 else 0)
            sent = (N + ring) * D * arr.itemsize
            old = N * G * D * arr.itemsize
            back = chip.read_rows([0, 1, N // 2, N - 1])
            assert back.tobytes() == db32[[0, 1, N // 2, N - 1]].tobytes()
            print(f"wire {wire} G={G}: {N} rows in {dt*1e3:7.1f} ms = {N/dt/1e6:5.2f} M rows/s; host->device bytes {sent/1e9:5.2f} GB "
                  f"({sent / (N * D * arr.itemsize):.2f}x the batch; round 2: {old/1e9:5.2f} GB = {G}x) -> {sent/dt/1e9:5.1f} GB/s", flush=True)
