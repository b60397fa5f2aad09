import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cerebro_amd import capi
D = 4096
for rows in (100_000, 1_000_000):
    with capi.Chip(D, capacity_hint=rows) as chip:
        chip.append_synthetic(rows, 1)
        for Q in (64, 128, 256, 512):
            q = chip.read_rows(np.arange(Q) * 37 % rows)
            chip.query_batch(rows, q, 8)
            chip.profile_enable(True); chip.profile_reset()
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                sc, ix = chip.query_batch(rows, q, 8)
            dt = (time.perf_counter() - t0) / n
            ms, cnt, b, span = chip.profile_scan()
            chip.profile_enable(False)
            kms = ms / cnt
            fl = 2.0 * Q * rows * D
            assert (ix[:, 0] == np.arange(Q) * 37 % rows).all()
            print(f"rows={rows} Q={Q}: call {dt*1e3:.2f} ms, kernel {kms:.2f} ms -> {fl/kms/1e9:.1f} TFLOP/s ({100*fl/kms/1e9/157.3:.1f}% of 157.3), {Q/dt:.0f} queries/s, HBM {rows*D*4*((Q+127)//128)/kms/1e6:.0f} GB/s")
