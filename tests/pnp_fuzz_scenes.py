"""The odd PnP scenes of the GPU fuzz (scripts/gpu_pnp_fuzz.py, tests/test_fuzz_gpu.py) -- data only.  Kept apart from the GPU script so
that tests/test_fuzz_scenes_reach_tie_paths.py can count, on the CPU, which of them reach the tie paths of the LU's pivot search."""
import numpy as np

import np_mirror_pnp as M


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
