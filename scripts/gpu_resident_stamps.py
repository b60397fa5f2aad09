"""Timeline of one command to the resident scan instance (CHIP_TICK_RESIDENT=1, CHIP_SCAN_STAMPS=1): s_memrealtime stamps (100 MHz) of
workgroup 0 seeing the command, its relay stores, every wave's entry / queries staged / rows done / end, the last workgroup's final
reduction, record and completion-word stores -- all on the device clock, relative to 'command seen'.  What the host adds on either side
is the synchronous tick (examples/sync_tick_latency.cc) minus this.   python scripts/gpu_resident_stamps.py [rows=10000]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
os.environ["CHIP_TICK_RESIDENT"] = "1"
os.environ["CHIP_SCAN_STAMPS"] = "1"
from cerebro_amd import capi  # noqa: E402

p = capi.default_dot_params()
p.min_new = -(1 << 30)
with capi.Chip(4096, capacity_hint=rows + 500) as chip:
    chip.append_synthetic(rows + 400, 1, [])
    fn = chip.lib.chip_debug_scan_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    grid, wpb = 256, 8
    nw = grid * wpb
    lines = []
    for rep in range(8):
        t0 = time.perf_counter()
        for i in range(40):
            chip.loop_tick(rows + 50 + 3 * (i % 100), p)
        dt = (time.perf_counter() - t0) / 40 * 1e6
        buf = np.zeros((nw + 2, 4), dtype=np.uint64)   # the waves' stamps + 8 pass-wide ones
        assert fn(chip.h, buf.ctypes.data, nw + 2) == 0          # (retires the instance; the next tick launches another)
        w = buf[:nw].astype(np.int64)
        x = buf[nw:].astype(np.int64).ravel()
        if rep < 2:
            continue
        us = lambda v: (v - x[0]) / 100.0
        ent, stg, rdone, wend = us(w[:, 0]), us(w[:, 1]), us(w[:, 2]), us(w[:, 3])
        lines.append([us(x[1]) if x[1] > 0 else np.nan, np.median(ent), ent.max(), np.median(stg), stg.max(), rdone.mean(), rdone.max(), wend.max(), us(x[5]), us(x[2]), us(x[6]), us(x[7]),
                      us(x[3]), us(x[4]), dt])
    a = np.array(lines)
    names = ["relay stores issued (nan: the host wrote every line)", "wave entry p50", "wave entry max", "queries staged p50", "queries staged max", "rows done mean", "rows done max",
             "wave end max", "last workgroup has the ticket", "its acquire fence done", "entries gathered + reduced", "barrier passed", "record stored",
             "completion word stored", "(python sync tick, us)"]
    print(f"rows={rows}  resident instance, {len(a)} samples (median over samples; us after workgroup 0 saw the command)")
    for n, col in zip(names, a.T):
        print(f"  {n:26s} {np.nanmedian(col):7.2f}   (min {np.nanmin(col):6.2f} max {np.nanmax(col):6.2f})" if not np.all(np.isnan(col)) else f"  {n:26s}     n/a")
    print(f"  ideal stream at 8 TB/s     {rows * 16384 / 8e6:7.2f}")
