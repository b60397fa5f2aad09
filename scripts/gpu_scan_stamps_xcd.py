#!/usr/bin/env python3
"""Where the tail of a SHORT scan launch comes from: per-wave wall-clock stamps of the row-batched kernel
(CHIP_SCAN_STAMPS=1) split into ramp / steady / tail and grouped by XCD (workgroup b runs on XCD b % 8), by workgroup slot and by
the number of rows a wave owns.   python scripts/gpu_scan_stamps_xcd.py [rows] [env assignments ...]"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

os.environ["CHIP_SCAN_STAMPS"] = "1"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 29_000
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
from cerebro_amd import capi  # noqa: E402

with capi.Chip(4096, capacity_hint=rows + 500) as chip:
    chip.append_synthetic(rows + 400, 1, [])
    fn = chip.lib.chip_debug_scan_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    nw = 512 * 16
    agg = []
    for rep in range(6):
        chip.loop_reset()
        chip.loop_tick(rows + 50 + 3 * rep)
        buf = np.zeros((nw, 4), dtype=np.uint64)
        assert fn(chip.h, buf.ctypes.data, nw) == 0
        if rep < 2:
            continue
        live = buf[:, 0] > 0
        t = buf.astype(np.int64)
        t0 = t[live, 0].min()
        us = (t - t0) / 100.0
        us[~live] = np.nan
        agg.append(us)
    us = np.nanmean(np.stack(agg), axis=0)       # mean over 4 launches, per wave slot
    live = ~np.isnan(us[:, 0])
    wpb = int(os.environ.get("CHIP_SCAN_BLOCK", "512")) // 64
    nwg = int(live.sum()) // wpb
    k = rows                                      # rows scanned ~ l - 50
    wave = np.arange(nw)
    wg = wave // wpb
    xcd = wg % 8
    n_rows_of = np.where(live, (k - 1 - wave) // int(live.sum()) + 1, 0)
    print(f"rows={rows} waves={int(live.sum())} workgroups={nwg}  (mean of 4 launches per wave slot)")
    print(f"  wave entry     p50 {np.nanmedian(us[:,0]):6.2f}  max {np.nanmax(us[:,0]):6.2f}")
    print(f"  staged         p50 {np.nanmedian(us[:,1]):6.2f}  max {np.nanmax(us[:,1]):6.2f}")
    print(f"  rows done      min {np.nanmin(us[:,2]):6.2f} p50 {np.nanmedian(us[:,2]):6.2f} mean {np.nanmean(us[:,2]):6.2f} p95 {np.nanpercentile(us[:,2],95):6.2f} max {np.nanmax(us[:,2]):6.2f}")
    print(f"  wave end       p50 {np.nanmedian(us[:,3]):6.2f}  max {np.nanmax(us[:,3]):6.2f}")
    ideal = rows * 16384 / 8e12 * 1e6
    print(f"  ideal stream at 8 TB/s {ideal:6.2f} us;  balanced finish (mean rows-done) {np.nanmean(us[:,2]):6.2f};  tail = max - mean = {np.nanmax(us[:,2]) - np.nanmean(us[:,2]):6.2f} us")
    print("  by XCD (rows done: mean / max):", "  ".join(f"{x}:{np.nanmean(us[(xcd==x)&live,2]):5.1f}/{np.nanmax(us[(xcd==x)&live,2]):5.1f}" for x in range(8)))
    for nr in sorted(set(n_rows_of[live].tolist())):
        m = live & (n_rows_of == nr)
        print(f"  waves owning {nr} rows: {int(m.sum()):5d}  rows done mean {np.nanmean(us[m,2]):6.2f} max {np.nanmax(us[m,2]):6.2f}")
    # first / second workgroup slot of a CU (blockIdx < 256 vs >= 256 is only a proxy: dispatch order)
    for lo, hi in ((0, nwg // 2), (nwg // 2, nwg)):
        m = live & (wg >= lo) & (wg < hi)
        print(f"  workgroups {lo:3d}..{hi-1:3d}: entry mean {np.nanmean(us[m,0]):5.2f}  rows done mean {np.nanmean(us[m,2]):6.2f} max {np.nanmax(us[m,2]):6.2f}")
    print("  by wave slot in its workgroup (rows done mean):", " ".join(f"{np.nanmean(us[live & (wave % wpb == i), 2]):5.1f}" for i in range(wpb)))
    # bandwidth timeline: bytes completed per 5 us bucket, assuming a wave's rows complete uniformly between staged and rows-done
    edges = np.arange(0, np.nanmax(us[:, 2]) + 5, 5.0)
    done = np.zeros(len(edges) - 1)
    for w in np.where(live)[0]:
        a, b, nr = us[w, 1], us[w, 2], n_rows_of[w]
        if b <= a:
            continue
        for i in range(len(done)):
            ov = max(0.0, min(b, edges[i + 1]) - max(a, edges[i]))
            done[i] += nr * 16384 * ov / (b - a)
    print("  TB/s per 5 us bucket:", " ".join(f"{d / 5e-6 / 1e12:4.1f}" for d in done))
