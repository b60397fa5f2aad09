"""Tiny driver for timeline traces of the pipelined tick loop: rows (default 10000) x 4096-D, 300 ticks, 16 in flight."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from cerebro_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
ls, plants, expect = bench.plan_ticks(rows, 320)
with capi.Chip(4096, capacity_hint=ls[-1]) as chip:
    chip.append_synthetic(ls[-1], bench.SEED, plants)
    p = capi.default_dot_params()
    bench.run_ticks(chip, ls[:20], p, 16)
    chip.synchronize()
    res = bench.run_ticks(chip, ls[20:], p, 16)
    chip.synchronize()
    bench.check_results(res, expect[20:])
print("ok")
