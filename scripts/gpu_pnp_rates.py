#!/usr/bin/env python3
"""PnP-RANSAC call times: 1000 hypotheses (single problem), reference mode (<= 50), batch of 8 problems."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cerebro_amd import capi
from cerebro_amd.synth import make_scene
X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
scenes = [make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242 + i)[:2] for i in range(8)]
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = 1000; p.seed = 4242
    for _ in range(3): chip.pnp_ransac(X, uv, p)
    t0 = time.perf_counter(); n = 40
    for i in range(n):
        p.seed = 4242 + i
        chip.pnp_ransac(X, uv, p)
    d1 = (time.perf_counter() - t0) / n
    p.n_hypotheses = 0
    t0 = time.perf_counter()
    for i in range(n): chip.pnp_ransac(X, uv, p)
    d0 = (time.perf_counter() - t0) / n
    p.n_hypotheses = 1000
    chip.pnp_ransac_batch(scenes, p)
    t0 = time.perf_counter(); m = 10
    for i in range(m): chip.pnp_ransac_batch(scenes, p, seeds=[1 + 8 * i + j for j in range(8)])
    d8 = (time.perf_counter() - t0) / m
print(f"1000 hyp {d1*1e6:.0f} us ({1000/d1/1e6:.3f} M hyp/s)   ref-mode {d0*1e6:.0f} us   batch8 {d8*1e6:.0f} us ({8000/d8/1e6:.3f} M hyp/s)", flush=True)
