#!/usr/bin/env python3
"""db_gemm_topk: TFLOP/s of the Q = 256 pass over 1M x 4096 + parity vs the fmaf oracle on a small DB."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cerebro_amd import capi
import oracle_lib, scenarios
rows, Q, D = 1_000_000, 256, 4096
small = scenarios.build_db(5, 3000, 1024, [])
for kc in (32,):
    for pd in (2, 4):   # stages of the LDS-DMA pipeline
        os.environ['CHIP_BATCH_STAGES'] = str(pd)
        with capi.Chip(1024) as c2:
            c2.append_f32(small)
            q = small[[5, 17, 2999, 1500, 0]]
            sc, ix = c2.query_batch(2950, q, 8)
            wsc, wix = oracle_lib.scan_topk_fmaf(small, 2950, q, 8)
            ok = np.array_equal(ix, wix) and np.array_equal(sc, wsc.astype(np.float32))
        with capi.Chip(D, capacity_hint=rows) as chip:
            chip.append_synthetic(rows, 1)
            qq = chip.read_rows((np.arange(Q) * 7919) % rows)
            chip.query_batch(rows, qq, 8)
            chip.profile_enable(True); chip.profile_reset()
            for _ in range(3): sc, ix = chip.query_batch(rows, qq, 8)
            ms, cnt, _, _ = chip.profile_scan()
            good = bool((ix[:, 0] == (np.arange(Q) * 7919) % rows).all())
        tf = 2.0 * Q * rows * D / (ms / 1e3 / cnt) / 1e12
        print(f"KC={kc} stages={pd}: {ms/cnt:.2f} ms  {tf:.1f} TFLOP/s ({tf/157.3:.3f})  parity={ok} self-hit={good}", flush=True)
