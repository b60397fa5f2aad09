#!/usr/bin/env python3
"""Round-5 A/B of the synchronous tick (one chip_loop_tick at a time, the live system's mode) over short prefixes: static row -> wave map
vs rows claimed within the workgroup (CHIP_SCAN_CLAIM), workgroup shapes, completion by polling (CHIP_TICK_POLL).  Every configuration is
a fresh ctx; all must return the same bits.   python scripts/gpu_sync_tick5.py [rows ...]"""
import json, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
from gpu_sync_tick import run

CONFIGS = [{}, {"CHIP_SCAN_CLAIM": "1"}] + [{"CHIP_SCAN_AGE_WEIGHTS": w} for w in
           ("0.2777,0.2585,0.2391,0.2248", "0.265,0.255,0.245,0.235", "0.29,0.26,0.235,0.215", "0.27,0.27,0.23,0.23", "0.285,0.255,0.24,0.22")]
rows_list = [int(x) for x in sys.argv[1:]] or [10_000, 29_000, 60_000]
for rows in rows_list:
    ref = None
    for env in CONFIGS:
        out, res = run(rows, env, n=300)
        if ref is None: ref = res
        out["same_bits"] = res == ref
        print(json.dumps(out), flush=True)
