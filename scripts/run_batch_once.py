"""Tiny driver for PMC passes on the batched MFMA kernel (rocprofv3 --pmc ...): one 256-query pass over 1M rows."""
import sys
sys.path.insert(0, '.')
import numpy as np
from cerebro_amd import capi
rows, Q, D = 1_000_000, 256, 4096
with capi.Chip(D, capacity_hint=rows) as chip:
    chip.append_synthetic(rows, 1)
    q = chip.read_rows((np.arange(Q) * 7919) % rows)
    for _ in range(3):
        sc, ix = chip.query_batch(rows, q, 8)
    print("ok", int((ix[:, 0] == (np.arange(Q) * 7919) % rows).sum()))
