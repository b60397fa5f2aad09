"""Throughput of chip_pnp_ransac_batch vs batch size (config 3 scene: N=512, H=1000)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import np_mirror_pnp as M
from cerebro_amd import capi
scenes = [M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242 + i)[:2] for i in range(16)]
with capi.Chip(64) as chip:
    for H in (0, 1000):
        for P in (1, 2, 4, 8, 16):
            p = capi.default_ransac_params(); p.n_hypotheses = H; p.seed = 4242
            for _ in range(3): chip.pnp_ransac_batch(scenes[:P], p)
            t0 = time.perf_counter(); n = 20
            for i in range(n): chip.pnp_ransac_batch(scenes[:P], p)
            dt = (time.perf_counter() - t0) / n
            hh = H if H else 50
            print(f"H={hh} P={P}: {dt*1e6:.0f} us per call, {P*hh/dt:.0f} hyp/s, {P/dt:.0f} problems/s")
