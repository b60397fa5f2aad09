"""GPU parity of the Umeyama-ICP-in-RANSAC leg (row N2): chip_icp_ransac vs oracle/icp_ransac.c, bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from cerebro_amd import capi
from test_oracle_icp import make_icp_scene

pytestmark = pytest.mark.gpu


def gparams(**kw):
    p = capi.default_icp_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def check(chip, A, B, **kw):
    o = O.icp_ransac(A, B, O.icp_params(**kw))
    g = chip.icp_ransac(A, B, gparams(**kw))
    for key in ("best_hypothesis", "n_iterations", "n_models", "n_inliers"):
        assert g["summary"][key] == o["summary"][key], (key, g["summary"], o["summary"])
    assert np.array_equal(g["mask"], o["mask"])
    if o["summary"]["best_hypothesis"] >= 0:
        assert np.linalg.norm(g["T"] - o["T"]) <= 1e-4 * np.linalg.norm(o["T"])
        assert np.array_equal(g["T"].view(np.uint64), o["T"].view(np.uint64))
        assert float(g["summary"]["best_cost"]).hex() == float(o["summary"]["best_cost"]).hex()
        assert g["confidence"] == o["confidence"]
    else:
        assert np.isnan(g["T"]).all() and g["confidence"] == 0.0
    return g, o


def test_defaults():
    p = capi.default_icp_params()
    assert (p.error_thresh, p.min_inlier_ratio, p.max_iterations, p.min_iterations, p.use_mle, p.sample_size) == (0.1, 0.7, 50, 5, 1, 10)


@pytest.mark.parametrize("N,outl,noise,seed", [(20, 0.0, 0.0, 1), (100, 0.1, 0.01, 2), (400, 0.25, 0.02, 11), (1000, 0.5, 0.05, 4), (3001, 0.3, 0.02, 5)])
def test_scenes_both_modes(N, outl, noise, seed):
    A, B, T, inl = make_icp_scene(N=N, outlier_frac=outl, noise=noise, seed=seed)
    with capi.Chip(64) as chip:
        g, o = check(chip, A, B, seed=seed)
        check(chip, A, B, seed=seed + 7, n_hypotheses=500)
        check(chip, A, B, seed=seed, use_mle=0, n_hypotheses=64)
        if outl <= 0.3:
            assert np.abs(g["T"][:3, :3] - T[:3, :3]).max() < 0.1


def test_scale_gate_and_edges():
    A, B, T, inl = make_icp_scene(N=200, outlier_frac=0.0, noise=0.0, seed=3)
    with capi.Chip(64) as chip:
        g, o = check(chip, A, 0.85 * B, seed=1)                   # every hypothesis fails min(s,1/s) > 0.9
        assert g["summary"]["best_hypothesis"] == -1
        check(chip, A, 1.05 * B, seed=1)
        r = chip.icp_ransac(A[:19], B[:19])
        assert r["status"] == capi.CHIP_ERR_TOO_FEW_POINTS and r["confidence"] == -1.0
        line = np.outer(np.arange(40.0), [1, 2, 3])
        check(chip, line, line + 1.0, seed=2)                      # collinear: no model
        Ap = A.copy(); Ap[:, 2] = 1.0                              # coplanar source points (rank-2 branch)
        Bp = Ap @ T[:3, :3].T + T[:3, 3]
        check(chip, Ap, Bp, seed=4, n_hypotheses=32)


def test_enqueue_collect_equals_the_blocking_call_and_overlaps_pnp():
    import np_mirror_pnp as M
    A, B, T, inl = make_icp_scene(N=400, outlier_frac=0.25, noise=0.02, seed=11)
    X, uv, Tp, _ = M.make_scene(N=300, outlier_frac=0.2, noise_px=0.5, seed=5)
    with capi.Chip(256) as chip:
        p = capi.default_icp_params(); p.seed = 77
        want = chip.icp_ransac(A, B, p)
        n = chip.icp_ransac_enqueue(A, B, p)
        with pytest.raises(capi.ChipError):                      # one estimation may be pending
            chip.icp_ransac_enqueue(A, B, p)
        pp = capi.default_ransac_params(); pp.seed = 9
        g = chip.pnp_ransac(X, uv, pp)                           # runs while the ICP kernel is in flight on its own stream
        got = chip.icp_ransac_collect(n)
        assert got["summary"] == want["summary"] and got["confidence"] == want["confidence"]
        assert np.array_equal(got["T"].view(np.uint64), want["T"].view(np.uint64)) and np.array_equal(got["mask"], want["mask"])
        o = O.pnp_ransac(X, uv, O.ransac_params(seed=9))
        assert np.array_equal(g["T"].view(np.uint64), o["T"].view(np.uint64))
        with pytest.raises(capi.ChipError):                      # nothing pending any more
            chip.icp_ransac_collect(n)
        assert chip.icp_ransac(A, B, p)["summary"] == want["summary"]


@pytest.mark.parametrize("N,outl,noise,seed", [(20, 0.0, 0.0, 1), (400, 0.25, 0.02, 11), (3001, 0.3, 0.02, 5)])
def test_theia_persistent_sampler_mode(N, outl, noise, seed):
    """The ICP estimator under theia::RandomSampler's persistent permutation (P3P_ICP builds one theia::Ransac per call,
    DlsPnpWithRansac.cpp:95-100): the host-sequenced sample table through icp_models, bit for bit the oracle's sampler mode 1."""
    A, B, T, inl = make_icp_scene(N=N, outlier_frac=outl, noise=noise, seed=seed)
    with capi.Chip(64) as chip:
        check(chip, A, B, seed=seed, sampler=capi.CHIP_SAMPLER_THEIA_PERSISTENT)
        check(chip, A, B, seed=seed + 3, n_hypotheses=300, sampler=1)
        check(chip, A, B, seed=seed + 3, n_hypotheses=300)
