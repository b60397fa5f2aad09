#!/usr/bin/env python3
"""Latency of ONE synchronous chip_loop_tick (the live system's mode: dot_product_th ticks at 10 Hz) over short prefixes, for the launch
shapes the library can choose from: CHIP_SCAN_SHORT_BPC (1 = half of every CU's workgroup slots, 0 = full grid), CHIP_SCAN_ROWS (rows per
wave in flight), CHIP_TICK_FUSED.  Every configuration is a fresh ctx; all must return the same bits."""
import json, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import bench
from cerebro_amd import capi

def run(rows, env, n=400):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ls, plants, expect = bench.plan_ticks(rows, 200)
        with capi.Chip(4096, capacity_hint=ls[-1]) as chip:
            chip.append_synthetic(ls[-1], bench.SEED, plants)
            p = capi.default_dot_params()
            p.min_new = -(1 << 30)
            for l in ls[:20]: chip.loop_tick(l, p)
            lat, res = [], []
            for i in range(n):
                l = ls[20 + i % 150]
                t = time.perf_counter(); r = chip.loop_tick(l, p); lat.append(time.perf_counter() - t)
                res.append((l, bytes(r)))
            lat = np.array(lat) * 1e6
            return dict(rows=rows, **env, mean_us=round(float(lat.mean()), 2), p50_us=round(float(np.median(lat)), 2), min_us=round(float(lat.min()), 2)), res
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

if __name__ == "__main__":
  for rows in (10_000, 29_000, 60_000, 100_000):
      ref = None
      for env in ({}, {"CHIP_SCAN_SHORT_BPC": "0"}, {"CHIP_SCAN_HALF_MIB": "100000"}, {"CHIP_SCAN_ROWS": "2"}, {"CHIP_SCAN_ROWS": "3"},
                  {"CHIP_SCAN_ROWS": "2", "CHIP_SCAN_SHORT_BPC": "0"}, {"CHIP_TICK_FUSED": "0"}, {"CHIP_SCAN_ROWS": "-1"}):
          out, res = run(rows, env)
          if ref is None: ref = res
          out["same_bits"] = res == ref
          print(json.dumps(out), flush=True)
