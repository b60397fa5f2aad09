"""Latency of the three estimations of one loop candidate (PNP a->b, PNP b->a, P3P_ICP; 512 correspondences, reference mode):
sequential blocking calls vs ICP enqueued underneath the batched PNP pair (what cerebro_hip::compute_three_way_pose does)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import np_mirror_pnp as M
from test_oracle_icp import make_icp_scene
from cerebro_amd import capi
X, uv, T, _ = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
X2, uv2, _, _ = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4243)
A, B = make_icp_scene(N=512, outlier_frac=0.3, noise=0.01, seed=7)[:2]
with capi.Chip(64) as chip:
    pp = capi.default_ransac_params(); pi = capi.default_icp_params()
    def seq():
        chip.pnp_ransac(X, uv, pp); chip.pnp_ransac(X2, uv2, pp); chip.icp_ransac(A, B, pi)
    def fused():
        n = chip.icp_ransac_enqueue(A, B, pi); chip.pnp_ransac_batch([(X, uv), (X2, uv2)], pp); chip.icp_ransac_collect(n)
    for name, f in (("sequential", seq), ("ICP underneath batched PNP pair", fused)):
        for _ in range(5): f()
        t0 = time.perf_counter(); n = 100
        for _ in range(n): f()
        print(f"three-way pose, {name}: {(time.perf_counter()-t0)/n*1e6:.0f} us per candidate")
