"""Tiny driver for PMC passes on the PnP-RANSAC kernels: config 3 (512 correspondences x 1000 hypotheses), 5 calls."""
import sys
sys.path.insert(0, '.')
from cerebro_amd import capi
from cerebro_amd.synth import make_scene
X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = 1000
    for i in range(5):
        p.seed = 4242 + i
        r = chip.pnp_ransac(X, uv, p)
    print("ok", r["summary"])
