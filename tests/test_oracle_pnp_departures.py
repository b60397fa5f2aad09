"""The PnP oracle DEFINES what Theia-SfM leaves unpinned ("parity unpinned").  These CPU tests put a
number on each documented departure, on the SURVEY 8d scene family, so that the "semantic parity" claim is measured:
  (1) root rule: oracle = real eigenpairs of the EISPACK QR only, complex pairs dropped outright; Theia (as recalled) = numpy-
      style eig, a root counts when |Im s| < 1e-6 (so a near-real conjugate pair counts TWICE) -- how often does
      "exactly one solution" (DlsPnpWithRansac.h:62) come out differently, and how far apart are the accepted poses?
  (2) sampler: oracle = fresh identity permutation per hypothesis; Theia = ONE index vector persisting across iterations --
      same marginal distribution of samples; how do the adaptive RANSAC outcomes (<= 50 iterations) compare?
  (3) Francis QR step: exact power-of-two scaling (frexp/ldexp) vs EISPACK's division by |p|+|q|+|r| (commit ada113f claimed
      "same hypotheses / masks"): both builds of the oracle over a fuzz corpus.
The measured numbers are printed (pytest -s) and quoted in DESIGN.md 6."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import np_mirror_pnp as M
import oracle_lib as O

ROOT = Path(__file__).resolve().parent.parent
SCENES = [dict(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242), dict(N=512, outlier_frac=0.1, noise_px=0.5, seed=7),
          dict(N=200, outlier_frac=0.2, noise_px=1.0, seed=11), dict(N=1000, outlier_frac=0.4, noise_px=0.3, seed=23)]


def rel_frob(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


def test_root_rule_departure_rate():
    n_hyp, disagree, both_one, pose_dev, near_real_pairs = 0, 0, 0, [], 0
    for sc in SCENES:
        X, uv, T, inl = M.make_scene(**sc)
        for hyp in range(120):
            smp = O.ransac_sample(sc["seed"], hyp, X.shape[0])
            u = O.dls_linear_form(sc["seed"], hyp)
            n, Rs, ts = O.dls_pnp(X[smp], uv[smp], u)
            sols, diag = M.dls_pnp(X[smp], uv[smp], u)          # numpy eig, |Im| < 1e-6 rule, cheirality on the sample
            n_hyp += 1
            s_all = [np.array([v[9], v[3], v[1]]) / v[0] for v in np.linalg.eig(diag["S"])[1].T]
            near_real_pairs += sum(1 for s in s_all if 0 < np.abs(s.imag).max() < 1e-6)
            if (n == 1) != (len(sols) == 1):
                disagree += 1
            elif n == 1:
                both_one += 1
                pose_dev.append(max(rel_frob(Rs[0], sols[0][0]), np.linalg.norm(ts[0] - sols[0][1]) / max(1e-12, np.linalg.norm(sols[0][1]))))
    print(f"\nroot rule: {n_hyp} hypotheses, accept/reject differs on {disagree} ({100 * disagree / n_hyp:.2f} %), both accept on {both_one}, "
          f"pose deviation between the two eigen-solvers: median {np.median(pose_dev):.2e}, 95 % {np.quantile(pose_dev, 0.95):.2e}, "
          f"max {max(pose_dev):.2e}; near-real complex roots seen: {near_real_pairs}")
    assert both_one >= 50
    assert disagree <= 0.01 * n_hyp                      # measured: 0 of 480
    # same root through two eigen-solvers (EISPACK-style QR here, LAPACK in numpy): the action matrix is non-normal, so a few
    # hypotheses are conditioned no better than ~1e-4 -- the scale of the north-star pose tolerance; typical agreement 1e-11
    assert np.median(pose_dev) < 1e-9 and max(pose_dev) < 1e-2   # measured: median 2e-11, max 2.2e-4


def _theia_style_run(X, uv, seed, persistent):
    """theia::Ransac::Estimate on the oracle's building blocks, with either sampler."""
    N, S = X.shape[0], 15
    idx = np.arange(N)
    prm = O.ransac_params()
    log_fail = np.log(prm.failure_probability)
    lib = O._bind_pnp()
    max_it = lib.orc_ransac_max_iterations(S, prm.min_inlier_ratio, log_fail, prm.min_iterations, prm.max_iterations)
    best_cost, best, it = np.inf, None, 0
    while it < max_it:
        if not persistent:
            idx = np.arange(N)
        for i in range(S):                                   # partial Fisher-Yates, RandInt(i, N-1) from the counter RNG
            j = i + int(lib.orc_rng_draw(seed, it, i) % (N - i))
            idx[i], idx[j] = idx[j], idx[i]
        smp = idx[:S].copy()
        n, Rs, ts = O.dls_pnp(X[smp], uv[smp], O.dls_linear_form(seed, it))
        it += 1
        if n != 1:
            continue
        T = np.eye(4); T[:3, :3] = Rs[0]; T[:3, 3] = ts[0]
        cost, nin, mask = O.score_model(T, X, uv, prm.error_thresh, prm.use_mle)
        if cost < best_cost:
            best_cost, best = cost, (T, nin, mask)
            ratio = nin / N
            if ratio >= S / N:
                max_it = min(max_it, lib.orc_ransac_max_iterations(S, ratio, log_fail, prm.min_iterations, prm.max_iterations))
    return best, it


def test_persistent_permutation_sampler_departure():
    rows = []
    for sc in (dict(N=512, outlier_frac=0.1, noise_px=0.5), dict(N=512, outlier_frac=0.3, noise_px=0.5)):
        for seed in range(100, 124):
            X, uv, T, inl = M.make_scene(seed=seed, **sc)
            fresh, it_f = _theia_style_run(X, uv, seed, persistent=False)
            pers, it_p = _theia_style_run(X, uv, seed, persistent=True)
            # the fresh-permutation python driver IS the oracle's driver
            o = O.pnp_ransac(X, uv, O.ransac_params(seed=seed))
            assert (fresh is None) == (o["summary"]["best_hypothesis"] < 0) and it_f == o["summary"]["n_iterations"]
            if fresh is not None:
                assert np.array_equal(fresh[2], o["mask"]) and rel_frob(fresh[0], o["T"]) < 1e-12
            # ... and the persistent-permutation python driver IS the oracle's sampler mode 1 (round 5: theia::RandomSampler as
            # written is a selectable mode of oracle and kernels, DlsPnpWithRansac.cpp:216-221)
            o1 = O.pnp_ransac(X, uv, O.ransac_params(seed=seed, sampler=1))
            assert (pers is None) == (o1["summary"]["best_hypothesis"] < 0) and it_p == o1["summary"]["n_iterations"]
            if pers is not None:
                assert np.array_equal(pers[2], o1["mask"]) and rel_frob(pers[0], o1["T"]) < 1e-12
            rows.append((fresh is not None, pers is not None, it_f, it_p,
                         rel_frob(fresh[0], T) if fresh else np.nan, rel_frob(pers[0], T) if pers else np.nan,
                         fresh[1] / X.shape[0] if fresh else 0.0, pers[1] / X.shape[0] if pers else 0.0))
    r = np.array(rows, dtype=float)
    ok_f, ok_p = r[:, 0].mean(), r[:, 1].mean()
    both = (r[:, 0] > 0) & (r[:, 1] > 0)
    print(f"\nsampler: success fresh {ok_f:.2f} / persistent {ok_p:.2f}; mean iterations {r[:, 2].mean():.1f} / {r[:, 3].mean():.1f}; "
          f"median pose error vs truth {np.nanmedian(r[:, 4]):.2e} / {np.nanmedian(r[:, 5]):.2e}; "
          f"median inlier ratio {np.median(r[both, 6]):.3f} / {np.median(r[both, 7]):.3f}")
    assert abs(ok_f - ok_p) <= 0.15 and min(ok_f, ok_p) >= 0.6
    assert 0.5 < np.nanmedian(r[:, 4]) / np.nanmedian(r[:, 5]) < 2.0
    assert abs(np.median(r[both, 6]) - np.median(r[both, 7])) < 0.03


def test_power_of_two_scaling_selects_like_eispack_division():
    so = ROOT / "oracle" / "_build" / "liboracle_eispack.so"
    if not so.exists():
        import subprocess
        subprocess.run(["make", "oracle"], cwd=ROOT, check=True, capture_output=True)
    e = C.CDLL(str(so))
    V = C.c_void_p
    e.orc_pnp_ransac.restype = C.c_int
    e.orc_pnp_ransac.argtypes = [V, V, C.c_int32, C.POINTER(O.OrcRansacParams), V, C.POINTER(C.c_float), V, C.POINTER(O.OrcRansacSummary)]
    worst, n_cases, n_models = 0.0, 0, 0
    for sc in SCENES:
        for seed in range(6):
            X, uv, T, inl = M.make_scene(**dict(sc, seed=sc["seed"] + seed))
            for nh in (0, 150):
                prm = O.ransac_params(n_hypotheses=nh, seed=seed + 1)
                a = O.pnp_ransac(X, uv, prm)
                Tb = np.empty(16); conf = C.c_float(); mask = np.zeros(X.shape[0], dtype=np.uint8); s = O.OrcRansacSummary()
                e.orc_pnp_ransac(X.ctypes.data_as(V), uv.ctypes.data_as(V), X.shape[0], C.byref(prm), Tb.ctypes.data_as(V), C.byref(conf),
                                 mask.ctypes.data_as(V), C.byref(s))
                n_cases += 1
                n_models += s.n_models
                assert (s.best_hypothesis, s.n_models, s.n_iterations, s.n_inliers) == \
                    (a["summary"]["best_hypothesis"], a["summary"]["n_models"], a["summary"]["n_iterations"], a["summary"]["n_inliers"])
                assert np.array_equal(mask, a["mask"])
                if s.best_hypothesis >= 0:
                    worst = max(worst, rel_frob(Tb.reshape(4, 4).T, a["T"]))
    print(f"\nfrexp vs division: {n_cases} RANSAC runs, {n_models} accepted models, identical hypotheses / masks / iteration counts, "
          f"max pose deviation {worst:.2e}")
    assert worst < 1e-6                                  # measured 3.2e-9
