"""CPU tests pinning oracle/icp_ransac.c (Umeyama-ICP in RANSAC, DlsPnpWithRansac.h:104-166, .cpp:16-122) against an
independent numpy Umeyama (np.linalg.svd) and the cited reference semantics.  PARITY UNPINNED for Theia internals."""
import numpy as np
import pytest

import np_mirror_pnp as M
import oracle_lib as O


def np_umeyama(a, b):
    """Umeyama 1991, unit weights: b ~ s R a + t."""
    ma, mb = a.mean(0), b.mean(0)
    da, db = a - ma, b - mb
    Sg = db.T @ da / len(a)
    U, D, Vt = np.linalg.svd(Sg)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (da ** 2).sum(1).mean()
    return R, mb - s * R @ ma, s


from cerebro_amd.synth import make_icp_scene  # noqa: E402,F401  (one definition, shared with bench.py's icp leg)


@pytest.mark.parametrize("seed", range(8))
def test_umeyama_matches_numpy_svd(seed):
    rng = np.random.default_rng(seed)
    A, B, T, inl = make_icp_scene(N=60, outlier_frac=0.0, noise=0.01 * (seed % 3), seed=seed)
    idx = rng.choice(60, 10, replace=False)
    scale = [1.0, 0.95, 1.3][seed % 3]
    b = B[idx] * scale
    rc, R, t, s = O.umeyama(A[idx], b)
    Rn, tn, sn = np_umeyama(A[idx], b)
    assert rc == 0
    assert np.abs(R - Rn).max() < 1e-9 and np.abs(t - tn).max() < 1e-8 and abs(s - sn) < 1e-10
    assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12


def test_umeyama_reflection_and_degenerate():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((10, 3))
    Rm = np.diag([1.0, 1.0, -1.0])                       # mirrored target: best PROPER rotation, S22 = -1 branch
    b = a @ Rm.T + 0.01 * rng.standard_normal((10, 3))
    rc, R, t, s = O.umeyama(a, b)
    Rn, tn, sn = np_umeyama(a, b)
    assert rc == 0 and np.linalg.det(R) > 0 and np.abs(R - Rn).max() < 1e-8 and abs(s - sn) < 1e-9
    ap = a.copy(); ap[:, 2] = 0.0                        # coplanar source: rank 2, still a rotation
    bp = ap @ M.quat_R([0.1, -0.2, 0.05]).T + 1.0
    rc, R, t, s = O.umeyama(ap, bp)
    assert rc == 0 and abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(ap @ R.T * s + t - bp).max() < 1e-8
    line = np.outer(np.arange(10.0), [1, 2, 3])          # collinear: rotation undetermined -> rejected
    assert O.umeyama(line, line + 1)[0] == -1


def test_hypothesis_scale_gate_and_error():
    A, B, T, inl = make_icp_scene(N=200, outlier_frac=0.0, noise=0.0, seed=3)
    ok, Th, s = O.icp_hypothesis(A, B, 7, 0)
    assert ok == 1 and abs(s - 1) < 1e-9 and np.abs(Th - T).max() < 1e-9
    ok, Th, s = O.icp_hypothesis(A, 0.85 * B, 7, 0)      # min(s, 1/s) = 0.85 < 0.9 -> rejected (DlsPnpWithRansac.h:137)
    assert ok == 0 and abs(s - 0.85) < 1e-9
    ok, Th, s = O.icp_hypothesis(A, 1.05 * B, 7, 0)      # 1/1.05 = 0.952 -> accepted, pose keeps R and t (not the scale)
    assert ok == 1 and abs(np.linalg.det(Th[:3, :3]) - 1) < 1e-12
    lib = O._bind_icp()
    Tc = np.ascontiguousarray(T.T.reshape(16))
    e = lib.orc_icp_error(Tc.ctypes.data, A[0].ctypes.data, (B[0] + [0.3, 0.0, 0.4]).ctypes.data)
    assert abs(e - 0.5) < 1e-12                          # L2 norm (:157)


def test_icp_ransac_scene_and_modes():
    A, B, T, inl = make_icp_scene(N=400, outlier_frac=0.25, noise=0.02, seed=11)
    p = O.icp_params()
    assert (p.error_thresh, p.min_inlier_ratio, p.max_iterations, p.min_iterations, p.use_mle, p.sample_size) == (0.1, 0.7, 50, 5, 1, 10)
    r = O.icp_ransac(A, B, O.icp_params(seed=3))
    assert r["status"] == 0 and 5 <= r["summary"]["n_iterations"] <= 50
    assert np.abs(r["T"][:3, :3] - T[:3, :3]).max() < 0.05 and np.abs(r["T"][:3, 3] - T[:3, 3]).max() < 0.2
    assert (r["mask"].astype(bool) & inl).sum() > 0.9 * inl.sum()
    rb = O.icp_ransac(A, B, O.icp_params(seed=3, n_hypotheses=300))
    assert rb["summary"]["n_iterations"] == 300 and rb["summary"]["best_cost"] <= r["summary"]["best_cost"] + 1e-12
    assert O.icp_ransac(A[:19], B[:19])["status"] == -9   # DlsPnpWithRansac.cpp:19-22
