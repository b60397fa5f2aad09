#!/usr/bin/env python3
"""Where the time of one SHORT scan launch goes (BASELINE config 2: 4096-D x 10k): per-wave wall-clock stamps of the row-batched kernel
(CHIP_SCAN_STAMPS=1: entry, queries staged + first loads issued, rows done, block merge done), one isolated launch at a time."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

os.environ["CHIP_SCAN_STAMPS"] = "1"
from cerebro_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
for R in (1, 2, 3):
    os.environ["CHIP_SCAN_ROWS"] = str(R)
    with capi.Chip(4096, capacity_hint=rows + 500) as chip:
        chip.append_synthetic(rows + 400, 1, [])
        fn = chip.lib.chip_debug_scan_stamps
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        nw = 512 * 8
        for rep in range(4):
            chip.loop_reset()
            chip.loop_tick(rows + 50 + 3 * rep)
            buf = np.zeros((nw, 4), dtype=np.uint64)
            assert fn(chip.h, buf.ctypes.data, nw) == 0
            live = buf[:, 0] > 0
            t = buf[live].astype(np.int64)
            t0 = t[:, 0].min()
            us = (t - t0) / 100.0          # s_memrealtime: 100 MHz
            if rep == 0:
                continue                   # first launch: cold caches, lazy code load
            q = lambda a: f"min {a.min():6.2f} p50 {np.median(a):6.2f} p95 {np.percentile(a, 95):6.2f} max {a.max():6.2f}"
            print(f"rows={rows} R={R} rep={rep} waves={live.sum()}:")
            print("   wave entry        ", q(us[:, 0]))
            print("   staged - entry    ", q(us[:, 1] - us[:, 0]))
            print("   rows done - staged", q(us[:, 2] - us[:, 1]))
            print("   merge done - rows ", q(us[:, 3] - us[:, 2]))
            print("   wave end          ", q(us[:, 3]), flush=True)
