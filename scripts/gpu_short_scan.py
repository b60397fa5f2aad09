#!/usr/bin/env python3
"""A/B of the tick over short and long prefixes (BASELINE configs 2, 3 and the headline): legacy one-row kernel vs the
row-batched kernel (db_scan_topk_rows, R = 1..3, temporal / non-temporal loads), ctx-stream vs same-stream merge.
Every configuration is a fresh ctx in this process (the CHIP_* knobs are read at chip_create); all must return the same bits.

  python scripts/gpu_short_scan.py [--rows 10000,100000] [--ticks 600] > gpurun_out/short_scan.txt
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

import bench  # noqa: E402
from cerebro_amd import capi  # noqa: E402

KNOBS = ("CHIP_TICK_FUSED", "CHIP_SCAN_ROWS", "CHIP_TICK_SAME_STREAM", "CHIP_SCAN_PLAIN_MIB", "CHIP_SCAN_VARIANT", "CHIP_SCAN_STREAMS", "CHIP_SCAN_HALF_MIB", "CHIP_SCAN_SHORT_BPC", "CHIP_SCAN_STREAMS3")


def run_config(rows, env, n_ticks, inflight):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ls, plants, expect = bench.plan_ticks(rows, n_ticks + 20)
    with capi.Chip(4096, capacity_hint=ls[-1]) as chip:
        chip.append_synthetic(ls[-1], bench.SEED, plants)
        params = capi.default_dot_params()
        chip.loop_reset()
        bench.run_ticks(chip, ls[:20], params, inflight)
        chip.synchronize()
        best = None
        for _ in range(3):
            chip.loop_reset()
            t0 = time.perf_counter()
            res = bench.run_ticks(chip, ls[20:], params, inflight)
            chip.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        bench.check_results(res, expect[20:])
        sig = [(r.found, r.idx_prev, list(r.argmax), [float(x).hex() for x in r.maxv]) for r in res[:40]]
        chip.loop_reset()
        chip.profile_enable(True)
        chip.profile_reset()
        bench.run_ticks(chip, ls[20:80], params, inflight)
        ms, cnt, _, _ = chip.profile_scan()
        chip.profile_enable(False)
        # latency of one synchronous tick (enqueue + collect, nothing in flight)
        chip.loop_reset()
        t0 = time.perf_counter()
        for l in ls[20:120]:
            chip.loop_tick(l, params)
        sync_us = (time.perf_counter() - t0) / 100 * 1e6
    n = len(ls) - 20
    return {"rows": rows, "env": env, "us_per_tick": best / n * 1e6, "ticks_per_s": n / best, "kernel_us_profiled": ms / max(cnt, 1) * 1e3,
            "sync_tick_us": sync_us, "GBps_step": 4.0 * 4096 * rows / (best / n) / 1e9}, sig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="10000,100000")
    ap.add_argument("--ticks", type=int, default=600)
    ap.add_argument("--inflight", type=int, default=16)
    ap.add_argument("--set", default="short", choices=["short", "long", "half"])
    args = ap.parse_args()
    short = [
        {"CHIP_SCAN_ROWS": -1, "CHIP_TICK_SAME_STREAM": 0},                                  # round-2 behaviour
        {"CHIP_SCAN_ROWS": -1, "CHIP_TICK_SAME_STREAM": 1},
        {"CHIP_SCAN_ROWS": 0, "CHIP_TICK_SAME_STREAM": 0},
        {"CHIP_SCAN_ROWS": 0, "CHIP_TICK_SAME_STREAM": 1},                                   # default
        {"CHIP_SCAN_ROWS": 0, "CHIP_TICK_SAME_STREAM": 1, "CHIP_SCAN_PLAIN_MIB": 0},         # nt loads on a cache-resident prefix
        {"CHIP_SCAN_ROWS": 1, "CHIP_TICK_SAME_STREAM": 1},
        {"CHIP_SCAN_ROWS": 2, "CHIP_TICK_SAME_STREAM": 1},
        {"CHIP_SCAN_ROWS": 3, "CHIP_TICK_SAME_STREAM": 1},
        {"CHIP_SCAN_ROWS": 0, "CHIP_TICK_SAME_STREAM": 1, "CHIP_SCAN_STREAMS": 1},
    ]
    long_ = [
        {"CHIP_SCAN_ROWS": -1},
        {"CHIP_SCAN_ROWS": 1},
        {"CHIP_SCAN_ROWS": 2},
        {"CHIP_SCAN_ROWS": 3},
        {"CHIP_SCAN_ROWS": -1, "CHIP_SCAN_VARIANT": 7},     # legacy kernel without the fp64 query staging
    ]
    half = [
        {"CHIP_SCAN_SHORT_BPC": 0},                                    # full-occupancy launches (call A's default)
        {"CHIP_SCAN_SHORT_BPC": 1},                                    # half-occupancy launches, two ticks resident together
        {"CHIP_SCAN_SHORT_BPC": 1, "CHIP_SCAN_ROWS": 1},
        {"CHIP_SCAN_SHORT_BPC": 1, "CHIP_SCAN_ROWS": 2},
        {"CHIP_SCAN_SHORT_BPC": 1, "CHIP_SCAN_PLAIN_MIB": 0},
        {"CHIP_SCAN_SHORT_BPC": 1, "CHIP_TICK_SAME_STREAM": 0},
        {"CHIP_SCAN_SHORT_BPC": 1, "CHIP_SCAN_ROWS": -1},              # legacy one-row kernel at half occupancy
    ]
    for rows in [int(x) for x in args.rows.split(",")]:
        ref = None
        for env in {"short": short, "long": long_, "half": half}[args.set]:
            try:
                r, sig = run_config(rows, env, args.ticks, args.inflight)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"rows": rows, "env": env, "error": repr(e)}), flush=True)
                continue
            if ref is None:
                ref = sig
            r["same_bits_as_first_config"] = sig == ref
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
