"""Tiny driver for kernel-trace passes on the PnP kernels in reference mode (<= 50 hypotheses): 20 calls; CHIP_PNP_DEBUG_STOP selects the stage."""
import sys
sys.path.insert(0, '.')
from cerebro_amd import capi
from cerebro_amd.synth import make_scene
X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 50
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = H
    for i in range(20):
        p.seed = 4242 + i
        r = chip.pnp_ransac(X, uv, p)
    print("ok", r["summary"])
