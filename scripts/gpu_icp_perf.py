"""Umeyama-ICP-RANSAC (row N2) call time: 512 3-D/3-D correspondences, reference mode (<= 50 iterations) and 1000 hypotheses."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_oracle_icp import make_icp_scene
from cerebro_amd import capi
A, B, T, inl = make_icp_scene(N=512, outlier_frac=0.3, noise=0.01, seed=7)[:4]
with capi.Chip(64) as chip:
    for H in (0, 1000, 8000):
        p = capi.default_icp_params(); p.n_hypotheses = H; p.seed = 3
        for _ in range(3): chip.icp_ransac(A, B, p)
        t0 = time.perf_counter(); n = 50
        for i in range(n):
            p.seed = 3 + i
            r = chip.icp_ransac(A, B, p)
        dt = (time.perf_counter() - t0) / n
        hh = H if H else 50
        print(f"ICP H={hh}: {dt*1e6:.0f} us per call, {hh/dt:.0f} hyp/s, models {r['summary']['n_models']}")
