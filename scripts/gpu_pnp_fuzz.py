"""Fuzz (also run, smaller, by tests/test_fuzz_gpu.py): GPU PnP/ICP RANSAC vs the oracle on many odd scenes (planar, duplicated points, tiny/huge scale, heavy
outliers, minimal N).  Everything must match bit for bit; prints a summary."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import np_mirror_pnp as M
import oracle_lib as O
from cerebro_amd import capi

def scene(i, rng):
    N = int(rng.choice([20, 21, 33, 64, 65, 100, 129, 200]))
    X, uv, T, inl = M.make_scene(N=N, outlier_frac=float(rng.choice([0, 0.1, 0.5, 0.9])), noise_px=float(rng.choice([0, 0.3, 3.0])), seed=1000 + i)
    kind = i % 12
    if kind == 1: X[:, 2] = X[:, 2].mean()                       # planar in depth (uv no longer consistent: mostly no model)
    if kind == 2: X[5:15] = X[5]; uv[5:15] = uv[5]               # duplicated correspondences
    if kind == 3: X *= 1e-3
    if kind == 4: X *= 1e3
    if kind == 5: uv[:] = uv[rng.permutation(N)]                 # total mismatch
    if kind == 6: X[:, 0] = 0.0                                  # points on a plane through the camera axis
    if kind == 7: X = np.round(X, 1); uv = np.round(uv, 2)       # coarse values
    # Symmetric scenes: the cubics' coefficients repeat, so the LU's pivot search meets EXACT ties and candidates that agree in their
    # high words (checked with an instrumented oracle: ties at elimination steps 2 .. 65; kinds 0-7 never produce one in 2.2 M
    # columns) -- the only inputs that reach the factor wave's tie path (64-bit compares, smallest logical row index, replayed swaps).
    if kind == 8: X[:, 1] = X[:, 0]; uv[:, 1] = uv[:, 0]         # mirror-symmetric in x <-> y
    if kind == 9: X[:, 1] = -X[:, 0]; uv[:, 1] = -uv[:, 0]
    if kind == 10: X[:, 2] = 4.0; uv[:, 0] = X[:, 0] / 4; uv[:, 1] = X[:, 1] / 4   # fronto-parallel plane seen from the identity pose, exact
    if kind == 11: X[:, 0] = 0.0; X[:, 1] = 0.0; uv[:] = 0.0     # every point on the optical axis: all-zero pivot columns, no model
    return X, uv

def same(g, o):
    ok = g["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"] and g["summary"]["n_models"] == o["summary"]["n_models"] \
        and g["summary"]["n_iterations"] == o["summary"]["n_iterations"] and np.array_equal(g["mask"], o["mask"])
    if o["summary"]["best_hypothesis"] >= 0:
        ok = ok and np.array_equal(g["T"].view(np.uint64), o["T"].view(np.uint64)) and g["confidence"] == o["confidence"]
    else:
        ok = ok and bool(np.isnan(g["T"]).all())
    return ok

def run(n=240, seed=7):
    rng = np.random.default_rng(seed)
    bad = []
    n_models = 0
    with capi.Chip(64) as chip:
        for i in range(n):
            X, uv = scene(i, rng)
            for H in (0, 60):
                sd = 5000 + i
                p = capi.default_ransac_params(); p.n_hypotheses = H; p.seed = sd
                g = chip.pnp_ransac(X, uv, p)
                o = O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=H, seed=sd))
                n_models += o["summary"]["n_models"]
                if not same(g, o): bad.append(("pnp", i, H, g["summary"], o["summary"]))
            A = X; B = X @ M.make_scene(N=20, seed=i)[2][:3, :3].T + rng.normal(0, 0.01, X.shape)
            pi = capi.default_icp_params(); pi.n_hypotheses = 40; pi.seed = 9000 + i
            gi = chip.icp_ransac(A, B, pi)
            oi = O.icp_ransac(A, B, O.icp_params(n_hypotheses=40, seed=9000 + i))
            if not same(gi, oi): bad.append(("icp", i, gi["summary"], oi["summary"]))
    return bad, n_models


if __name__ == "__main__":
    t0 = time.time()
    bad, n_models = run(int(sys.argv[1]) if len(sys.argv) > 1 else 240)
    print(f"fuzz: {len(bad)} mismatches, {n_models} oracle models, {time.time()-t0:.1f} s")
    for b in bad[:10]: print(b)
