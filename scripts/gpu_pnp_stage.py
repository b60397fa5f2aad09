import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import np_mirror_pnp as M
from cerebro_amd import capi
capi.use_hooks_library().__enter__()   # CHIP_PNP_DEBUG_STOP exists in the TEST build of the library only
X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
for H in (50, 1000):
    for stop in (1, 2, 3, 4, 0):
        os.environ['CHIP_PNP_DEBUG_STOP'] = str(stop)
        with capi.Chip(64) as chip:
            p = capi.default_ransac_params(); p.n_hypotheses = H; p.seed = 4242
            for _ in range(3): chip.pnp_ransac(X, uv, p)
            t0 = time.perf_counter(); n = 30
            for i in range(n): chip.pnp_ransac(X, uv, p)
            dt = (time.perf_counter() - t0) / n
        print(f"H={H} stop={stop}: {dt*1e6:.0f} us per call")
