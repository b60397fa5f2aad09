"""Host-side time per tick on the single-GPU pipelined path (enqueue / collect) vs DB size."""
import sys, time
sys.path.insert(0, '.')
import torch
from cerebro_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chip = capi.Chip(4096, capacity_hint=rows + 2000)
chip.append_synthetic(rows + 1000, 1, [])
params = capi.default_dot_params()
T = {"enqueue": 0.0, "collect": 0.0}
def run(n, l0):
    pend = []
    for i in range(n):
        l = l0 + 3 * i
        if len(pend) == W:
            t = time.perf_counter(); chip.loop_tick_collect(pend.pop(0)); T["collect"] += time.perf_counter() - t
        s = i % W
        t = time.perf_counter(); chip.loop_tick_enqueue(l, s, params); T["enqueue"] += time.perf_counter() - t
        pend.append(s)
    while pend: chip.loop_tick_collect(pend.pop(0))
chip.loop_reset(); run(20, rows - 200); chip.synchronize()
for k in T: T[k] = 0.0
chip.loop_reset(); t0 = time.perf_counter(); n = 300; run(n, rows - 950); chip.synchronize(); dt = time.perf_counter() - t0
print(f"rows={rows} W={W}: {dt/n*1e6:.1f} us/tick;", {k: round(v / n * 1e6, 1) for k, v in T.items()})
chip.close()
