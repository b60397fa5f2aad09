"""Tiny driver for rocprofv3 traces / PMC passes of the tick loop: `rows` (default 10000) x DIM-D, float or double rows.
  python scripts/run_ticks_once.py ROWS                          300 pipelined ticks, 16 in flight (launches of consecutive ticks overlap)
  python scripts/run_ticks_once.py ROWS sync [N] [DIM] [f32|f64] N (default 60) SYNCHRONOUS ticks, one launch at a time -- per-launch durations and
                                                                 counters of the isolated kernel (profiles/scan_traffic_sizes.json)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from cerebro_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
sync = len(sys.argv) > 2 and sys.argv[2] == "sync"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
storage = sys.argv[5] if len(sys.argv) > 5 else "f32"
ls, plants, expect = bench.plan_ticks(rows, 320)
with capi.Chip(dim, capacity_hint=ls[-1], storage=(None if storage == "f32" else "f64")) as chip:
    chip.append_synthetic(ls[-1], bench.SEED, plants)
    p = capi.default_dot_params()
    bench.run_ticks(chip, ls[:20], p, 16)
    chip.synchronize()
    if sync:
        res = [chip.loop_tick(l, p) for l in ls[20:20 + n]]
        bench.check_results(res, expect[20:20 + n])
    else:
        res = bench.run_ticks(chip, ls[20:], p, 16)
        chip.synchronize()
        bench.check_results(res, expect[20:])
print("ok")
