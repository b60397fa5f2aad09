"""CPU tests pinning oracle/pnp_ransac.c.  The reference holds no vectors for this path and Theia is absent
(PARITY UNPINNED for Theia internals), so the oracle is validated by (a) the cited reference semantics,
(b) self-consistency on noise-free data, (c) an independent numpy DLS implementation (np.linalg.eig/solve),
(d) RANSAC behaviour on the SURVEY 8d scene, (e) a committed golden fixture of its own outputs."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

import np_mirror_pnp as M
import oracle_lib as O

GOLD = Path(__file__).parent / "golden"


def rel_frob(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


def test_params_match_reference():
    p = O.ransac_params()
    # DlsPnpWithRansac.cpp:207-212, DlsPnpWithRansac.h:45
    assert (p.error_thresh, p.min_inlier_ratio, p.max_iterations, p.min_iterations, p.use_mle, p.sample_size) == \
        (0.03, 0.7, 50, 5, 1, 15)
    assert p.failure_probability == 0.01


def test_max_iterations_worked_numbers():
    """S=15 -> ratio .7: 968 (clamped to 50), .9: 20, .95: 8, >=.97: 5."""
    lib = O._bind_pnp()
    lf = math.log(0.01)
    assert lib.orc_ransac_max_iterations(15, 0.7, lf, 5, 100000) == 968
    assert lib.orc_ransac_max_iterations(15, 0.7, lf, 5, 50) == 50
    assert lib.orc_ransac_max_iterations(15, 0.9, lf, 5, 50) == 20
    assert lib.orc_ransac_max_iterations(15, 0.95, lf, 5, 50) == 8
    assert lib.orc_ransac_max_iterations(15, 0.97, lf, 5, 50) == 5
    assert lib.orc_ransac_max_iterations(15, 1.0, lf, 5, 50) == 5


def test_sampler_is_partial_fisher_yates():
    N, S, seed = 512, 15, 99
    lib = O._bind_pnp()
    for hyp in (0, 1, 7, 999):
        got = O.ransac_sample(seed, hyp, N, S)
        idx = list(range(N))
        for i in range(S):
            j = i + lib.orc_rng_draw(seed, hyp, i) % (N - i)
            idx[i], idx[j] = idx[j], idx[i]
        assert list(got) == idx[:S]
        assert len(set(got)) == S and min(got) >= 0 and max(got) < N
    # all indices reachable, roughly uniform
    cnt = np.zeros(40)
    for hyp in range(4000):
        cnt[O.ransac_sample(5, hyp, 40, 15)] += 1
    assert cnt.min() > 0.8 * 4000 * 15 / 40 and cnt.max() < 1.2 * 4000 * 15 / 40
    u = O.dls_linear_form(5, 3)
    assert np.all(np.abs(u) < 100) and len(set(u)) == 4


def test_reproj_error_definition():
    """DlsPnpWithRansac.h:75-99: L1 error in normalized coordinates, no cheirality rejection."""
    X, uv, T, inl = M.make_scene(N=64, outlier_frac=0.0, noise_px=0.0, seed=3)
    cost, nin, mask = O.score_model(T, X, uv)
    assert nin == 64 and cost < 1e-6 and mask.all()
    P = X @ T[:3, :3].T + T[:3, 3]
    want = np.abs(P[:, 0] / P[:, 2] - 0.1 - uv[:, 0] + 0.1) + np.abs(P[:, 1] / P[:, 2] - uv[:, 1])
    lib = O._bind_pnp()
    Tc = np.ascontiguousarray(T.T.reshape(16))
    for i in range(5):
        e = lib.orc_reproj_error(Tc.ctypes.data, X[i].ctypes.data, uv[i].ctypes.data)
        assert abs(e - want[i]) < 1e-15
    # a point behind the camera is scored like any other (reference does not reject it)
    Xn = X.copy(); Xn[0] = -Xn[0]
    _, _, m2 = O.score_model(T, Xn, uv)
    assert m2[0] in (0, 1)
    # MLE cost = sum(min(r, thresh)); strict '<' for inliers
    uvb = uv.copy(); uvb[:10] += 1.0
    cost, nin, mask = O.score_model(T, X, uvb)
    assert nin == 54 and not mask[:10].any()
    assert cost == pytest.approx(10 * 0.03, abs=1e-6)
    cost0, _, _ = O.score_model(T, X, uvb, use_mle=0)
    assert cost0 == 10.0


@pytest.mark.parametrize("seed", range(6))
def test_dls_noise_free_recovers_pose(seed):
    X, uv, T, _ = M.make_scene(N=256, outlier_frac=0.0, noise_px=0.0, seed=100 + seed)
    idx = np.random.default_rng(seed).choice(256, 15, replace=False)
    n, Rs, ts = O.dls_pnp(X[idx], uv[idx], O.dls_linear_form(seed, 0))
    assert n >= 1
    errs = [rel_frob(Rs[i], T[:3, :3]) + np.linalg.norm(ts[i] - T[:3, 3]) for i in range(len(Rs))]
    assert min(errs) < 1e-8


@pytest.mark.parametrize("seed", range(12))
def test_dls_matches_numpy_mirror(seed):
    """Same sample, same linear form -> same solution set as the independent numpy implementation."""
    rng = np.random.default_rng(seed)
    X, uv, T, inl = M.make_scene(N=200, outlier_frac=0.25 if seed % 3 == 0 else 0.0, noise_px=0.5, seed=seed)
    idx = rng.choice(200, 15, replace=False)
    u = O.dls_linear_form(seed, 1)
    Tf, f = O.dls_cubics(X[idx], uv[idx])
    sols, dbg = M.dls_pnp(X[idx], uv[idx], u)
    mons = [(a, b, c) for a in range(4) for b in range(4 - a) for c in range(4 - a - b)]
    fm = np.array([[dbg["f"][k].get(m, 0.0) for m in mons] for k in range(3)])
    assert np.abs(f - fm).max() <= 1e-12 * np.abs(fm).max()
    assert np.abs(Tf - dbg["T"]).max() < 1e-10
    rc, S = O.dls_action_matrix(f, u)
    assert rc == 0 and np.abs(S - dbg["S"]).max() <= 1e-7 * np.abs(dbg["S"]).max()
    nr, lam, v4 = O.eig27_real(S)
    w = np.linalg.eigvals(S)
    wreal = np.sort(w[np.abs(w.imag) < 1e-8 * max(1.0, np.abs(w).max())].real)
    assert nr == len(wreal)
    assert np.allclose(np.sort(lam), wreal, rtol=1e-6, atol=1e-6)
    n, Rs, ts = O.dls_pnp(X[idx], uv[idx], u)
    assert n == len(sols)
    for i in range(min(n, len(Rs))):
        d = min(rel_frob(Rs[i], R) + np.linalg.norm(ts[i] - t) for R, t in sols)
        assert d < 1e-6


def test_hypothesis_accepts_iff_exactly_one_solution():
    """DlsPnpWithRansac.h:62: rots.size()==1.  Clean samples mostly give one solution, contaminated ones mostly none."""
    X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    ok_clean = ok_dirty = n_clean = n_dirty = 0
    for hyp in range(300):
        ok, Th, smp = O.pnp_hypothesis(X, uv, 77, hyp)
        assert list(smp) == list(O.ransac_sample(77, hyp, 512, 15))
        n, Rs, ts = O.dls_pnp(X[smp], uv[smp], O.dls_linear_form(77, hyp))
        assert ok == (1 if n == 1 else 0)
        if ok:
            assert np.allclose(Th[:3, :3], Rs[0]) and np.allclose(Th[:3, 3], ts[0]) and np.allclose(Th[3], [0, 0, 0, 1])
            assert abs(np.linalg.det(Th[:3, :3]) - 1) < 1e-12
        if inl[smp].all():
            n_clean += 1; ok_clean += ok
        else:
            n_dirty += 1; ok_dirty += ok
    assert n_dirty > 250                       # 0.7^15 = 0.5 % all-inlier samples
    assert ok_dirty / n_dirty < 0.5


def test_ransac_scene_benchmark_mode():
    X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    r = O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=1000, seed=4242))
    s = r["summary"]
    assert r["status"] == 0 and s["n_iterations"] == 1000 and s["best_hypothesis"] >= 0 and s["n_models"] > 50
    assert rel_frob(r["T"][:3, :3], T[:3, :3]) < 0.02 and np.linalg.norm(r["T"][:3, 3] - T[:3, 3]) < 0.2
    assert s["n_inliers"] == r["mask"].sum() and s["n_inliers"] > 0.9 * inl.sum()
    assert (r["mask"].astype(bool) & ~inl).sum() < 0.1 * (~inl).sum() + 5
    # the reported model is hypothesis `best_hypothesis`, re-scored
    ok, Th, _ = O.pnp_hypothesis(X, uv, 4242, s["best_hypothesis"])
    assert ok and np.array_equal(Th, r["T"])
    cost, nin, mask = O.score_model(Th, X, uv)
    assert cost == s["best_cost"] and nin == s["n_inliers"] and np.array_equal(mask, r["mask"])
    # argmin with lowest index winning ties == sequential strict '<'
    costs = []
    for h in range(1000):
        ok, Th, _ = O.pnp_hypothesis(X, uv, 4242, h)
        costs.append(O.score_model(Th, X, uv)[0] if ok else np.inf)
    assert int(np.argmin(costs)) == s["best_hypothesis"]
    conf = 1 - (1 - (s["n_inliers"] / 512) ** 15) ** 1000
    assert r["confidence"] == pytest.approx(np.float32(conf))


def test_ransac_reference_mode_adaptive_termination():
    X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.05, noise_px=0.3, seed=7)
    r = O.pnp_ransac(X, uv, O.ransac_params(seed=11))
    s = r["summary"]
    assert 5 <= s["n_iterations"] <= 50                      # min/max_iterations (DlsPnpWithRansac.cpp:210-211)
    assert rel_frob(r["T"][:3, :3], T[:3, :3]) < 0.02
    # replay the sequential rule of theia::Ransac
    lib = O._bind_pnp()
    best, max_it, it, best_h = np.inf, 50, 0, -1
    while it < max_it:
        ok, Th, _ = O.pnp_hypothesis(X, uv, 11, it)
        if ok:
            cost, nin, _ = O.score_model(Th, X, uv)
            if cost < best:
                best, best_h = cost, it
                if nin / 512 >= 15 / 512:
                    max_it = min(max_it, lib.orc_ransac_max_iterations(15, nin / 512, math.log(0.01), 5, 50))
        it += 1
    assert (it, best_h) == (s["n_iterations"], s["best_hypothesis"])
    conf = 1 - (1 - (s["n_inliers"] / 512) ** 15) ** it
    assert r["confidence"] == pytest.approx(np.float32(conf))


@pytest.mark.parametrize("thresh,ratio,cite", [(0.02, 0.90, "unittest_theia.cpp:489-494"), (0.02, 0.7, "unittest_theia.cpp:1283-1287, Cerebro.cpp:1976-1980")])
def test_ransac_with_the_reference_s_other_parameter_sets(thresh, ratio, cite):
    """Besides the production set (0.03 / 0.7, DlsPnpWithRansac.cpp:207-212) the reference runs DlsPnpWithRansac with two more:
    its own manual test uses error_thresh 0.02 with min_inlier_ratio 0.90 -- which caps the run at ComputeMaxIterations(15, .9) = 20
    iterations before the first hypothesis -- and a second (dead) copy uses 0.02 / 0.7.  The sequential rule replayed in Python."""
    X, uv, T, inl = M.make_scene(N=300, outlier_frac=0.04, noise_px=0.3, seed=21)
    prm = O.ransac_params(seed=5, error_thresh=thresh, min_inlier_ratio=ratio)
    r = O.pnp_ransac(X, uv, prm)
    s = r["summary"]
    lib = O._bind_pnp()
    max_it = min(50, lib.orc_ransac_max_iterations(15, ratio, math.log(0.01), 5, 50))
    assert max_it == (20 if ratio == 0.90 else 50), cite
    best, it, best_h = np.inf, 0, -1
    while it < max_it:
        ok, Th, _ = O.pnp_hypothesis(X, uv, 5, it)
        if ok:
            cost, nin, _ = O.score_model(Th, X, uv, thresh=thresh)
            if cost < best:
                best, best_h = cost, it
                if nin / 300 >= 15 / 300:
                    max_it = min(max_it, lib.orc_ransac_max_iterations(15, nin / 300, math.log(0.01), 5, 50))
        it += 1
    assert (it, best_h) == (s["n_iterations"], s["best_hypothesis"]) and 5 <= it <= (20 if ratio == 0.90 else 50)
    assert rel_frob(r["T"][:3, :3], T[:3, :3]) < 0.02


def test_ransac_edge_cases():
    X, uv, T, inl = M.make_scene(N=64, outlier_frac=0.0, noise_px=0.0, seed=1)
    assert O.pnp_ransac(X[:19], uv[:19])["status"] == -9            # DlsPnpWithRansac.cpp:136-139 (<20 points -> -1)
    assert O.pnp_ransac(X[:20], uv[:20])["status"] == 0
    # garbage correspondences: no hypothesis yields a model -> NaN pose (caller's NaN gate, Cerebro.cpp:1678)
    rng = np.random.default_rng(0)
    Xg = rng.uniform(-1, 1, (40, 3)); Xg[:, 2] = -np.abs(Xg[:, 2]) - 1.0
    r = O.pnp_ransac(Xg, rng.uniform(-1, 1, (40, 2)), O.ransac_params(n_hypotheses=20))
    if r["summary"]["best_hypothesis"] < 0:
        assert np.isnan(r["T"]).all() and r["confidence"] == 0 and r["mask"].sum() == 0


def test_golden_fixture_pnp():
    g = json.loads((GOLD / "pnp_golden.json").read_text())
    X = np.array(g["X"]); uv = np.array(g["uv"])
    for case in g["cases"]:
        r = O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=case["n_hypotheses"], seed=case["seed"]))
        assert r["summary"]["best_hypothesis"] == case["best_hypothesis"]
        assert r["summary"]["n_iterations"] == case["n_iterations"]
        assert r["summary"]["n_models"] == case["n_models"]
        assert np.packbits(r["mask"]).tobytes().hex() == case["mask_hex"]
        assert [float(x).hex() for x in r["T"].T.reshape(16)] == case["T_colmajor_hex"]


def test_all_cores_hypothesis_loop_equals_serial_driver():
    """bench.py's all-cores PnP baseline (OpenMP over hypotheses) selects the same benchmark-mode winner and counts the same
    number of models as the serial orc_pnp_ransac."""
    X, uv, T, inl = M.make_scene(N=128, outlier_frac=0.3, noise_px=0.5, seed=11)
    prm = O.ransac_params(n_hypotheses=120, seed=5)
    r = O.pnp_ransac(X, uv, prm)
    best, nm = O.pnp_hypotheses_mt(X, uv, prm, 120, 4)
    assert best == r["summary"]["best_hypothesis"] and nm == r["summary"]["n_models"]


def _dls_cost(R, X, uv):
    """Hesch & Roumeliotis' objective for a rotation R with the translation eliminated in closed form:
    J(R) = sum_i || (I - z_i z_i^T) (R X_i + t*(R)) ||^2, z_i = unit bearing of uv_i."""
    z = np.column_stack([uv, np.ones(len(uv))]); z /= np.linalg.norm(z, axis=1, keepdims=True)
    P = np.eye(3)[None] - z[:, :, None] * z[:, None, :]                     # (n,3,3) projectors orthogonal to the bearings
    RX = X @ R.T
    t = -np.linalg.solve(P.sum(0), np.einsum("nij,nj->i", P, RX))
    r = np.einsum("nij,nj->ni", P, RX + t)
    return float((r * r).sum()), t


def _rotvec(w):
    th = np.linalg.norm(w)
    if th == 0:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _cayley_R(s):
    """R(s) = ((1 - s.s) I + 2 [s]x + 2 s s^T) / (1 + s.s)  (Hesch & Roumeliotis eq. 9)"""
    K = np.array([[0, -s[2], s[1]], [s[2], 0, -s[0]], [-s[1], s[0], 0]])
    return ((1 - s @ s) * np.eye(3) + 2 * K + 2 * np.outer(s, s)) / (1 + s @ s)


def _cayley_s(R):
    """inverse of _cayley_R (rotation angle != pi): s = axis * tan(angle/2), read off the skew part: R - R^T = 4 [s]x / (1 + s.s)"""
    w = 0.5 * np.sqrt(max(1e-300, 1.0 + np.trace(R)))          # quaternion scalar part = cos(angle/2)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4.0 * w)
    return v / w


@pytest.mark.parametrize("seed", range(8))
def test_dls_solutions_are_stationary_points_of_the_least_squares_cost(seed):
    """Independent of any implementation detail: what the Macaulay/eigen machinery solves on NOISY data.  DLS minimises the
    Cayley-polynomial cost J'(s) = (1 + s.s)^2 J(R(s)) (the denominators of R(s) are cleared so that the optimality
    conditions are cubic); every returned rotation must be a stationary point of J' in s, with the returned translation
    being the closed-form minimiser for that rotation."""
    rng = np.random.default_rng(seed)
    X, uv, T, _ = M.make_scene(N=200, outlier_frac=0.0, noise_px=2.0, seed=300 + seed)
    idx = rng.choice(200, 15, replace=False)
    Xs, uvs = X[idx], uv[idx]
    n, Rs, ts = O.dls_pnp(Xs, uvs, O.dls_linear_form(seed, 2))
    assert n >= 1
    Jp = lambda s: (1 + s @ s) ** 2 * _dls_cost(_cayley_R(s), Xs, uvs)[0]
    for R, t in zip(Rs, ts):
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and np.linalg.det(R) > 0
        J0, t_star = _dls_cost(R, Xs, uvs)
        assert np.allclose(t, t_star, rtol=1e-7, atol=1e-9)                  # t = closed-form translation for that R
        s0 = _cayley_s(R)
        assert np.allclose(_cayley_R(s0), R, atol=1e-9)
        h = 1e-5
        g = np.array([(Jp(s0 + h * e) - Jp(s0 - h * e)) / (2 * h) for e in np.eye(3)])
        curv = max(abs(Jp(s0 + 1e-2 * e) + Jp(s0 - 1e-2 * e) - 2 * Jp(s0)) / 1e-4 for e in np.eye(3))   # second-derivative scale
        assert np.abs(g).max() <= 1e-6 * max(curv, 1e-12), (g, curv)          # gradient vanishes relative to the curvature


def test_persistent_sampler_is_theias_random_sampler():
    """Sampler mode 1 = theia::RandomSampler as written: Initialize() fills 0..N-1 ONCE; every Sample() does, for i < S,
    swap(idx[i], idx[RandInt(i, N-1)]) on the array the previous call left and returns idx[:S].  Restated here in plain Python over
    the same counter-based draws; hypothesis 0 coincides with the fresh-permutation mode, later ones in general do not."""
    lib = O._bind_pnp()
    for seed, N, S, H in ((7, 40, 15, 60), (1234567, 512, 15, 50), (3, 20, 10, 30), (99, 16, 15, 8)):
        idx = list(range(N))
        want = []
        for h in range(H):
            for i in range(S):
                j = i + int(lib.orc_rng_draw(seed, h, i) % (N - i))
                idx[i], idx[j] = idx[j], idx[i]
            want.append(idx[:S])
        got = O.ransac_sample_persistent(seed, H, N, S)
        assert got.tolist() == want
        assert got[0].tolist() == O.ransac_sample(seed, 0, N, S).tolist()
        assert all(len(set(r)) == S and min(r) >= 0 and max(r) < N for r in want)
        if N > 2 * S:
            assert any(got[h].tolist() != O.ransac_sample(seed, h, N, S).tolist() for h in range(1, H))
