"""Tiny driver for kernel traces of the batched PnP call: 8 problems x 1000 hypotheses per launch pair, 5 calls."""
import sys
sys.path.insert(0, '.')
from cerebro_amd import capi
from cerebro_amd.synth import make_scene
scenes = [make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242 + i)[:2] for i in range(8)]
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = 1000
    for i in range(5):
        chip.pnp_ransac_batch(scenes, p, seeds=[1 + 8 * i + j for j in range(8)])
print("ok")
