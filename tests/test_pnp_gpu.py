"""GPU parity tests of the DLS-PnP-in-RANSAC path: HIP kernels (through chip_pnp_ransac) vs the CPU oracle on
identical inputs and seeds.  Bar (BASELINE north_star): inlier masks bit-exact, poses within 1e-4 relative
Frobenius; in practice the kernels reproduce the oracle's operation order, so poses are compared bit for bit too."""
import json
from pathlib import Path

import numpy as np
import pytest

import np_mirror_pnp as M
import oracle_lib as O
from cerebro_amd import capi

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def rel_frob(A, B):
    return np.linalg.norm(A - B) / np.linalg.norm(B)


def gparams(**kw):
    p = capi.default_ransac_params()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def check_against_oracle(chip, X, uv, **kw):
    o = O.pnp_ransac(X, uv, O.ransac_params(**kw))
    g = chip.pnp_ransac(X, uv, gparams(**kw))
    assert g["status"] == 0 and o["status"] == 0
    assert g["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"], (g["summary"], o["summary"])
    assert g["summary"]["n_iterations"] == o["summary"]["n_iterations"]
    assert g["summary"]["n_models"] == o["summary"]["n_models"]
    assert g["summary"]["n_inliers"] == o["summary"]["n_inliers"]
    assert np.array_equal(g["mask"], o["mask"])                                   # inlier mask bit-exact
    if o["summary"]["best_hypothesis"] >= 0:
        assert rel_frob(g["T"], o["T"]) <= 1e-4                                    # north_star tolerance
        assert np.array_equal(g["T"].view(np.uint64), o["T"].view(np.uint64))     # and in fact bit-identical
        assert float(g["summary"]["best_cost"]).hex() == float(o["summary"]["best_cost"]).hex()
        assert g["confidence"] == o["confidence"]
    else:
        assert np.isnan(g["T"]).all() and g["confidence"] == 0.0 and g["mask"].sum() == 0
    return g, o


def test_golden_fixture_through_c_abi():
    g = json.loads((GOLD / "pnp_golden.json").read_text())
    X = np.array(g["X"]); uv = np.array(g["uv"])
    with capi.Chip(256) as chip:
        for case in g["cases"]:
            r = chip.pnp_ransac(X, uv, gparams(n_hypotheses=case["n_hypotheses"], seed=case["seed"]))
            assert r["summary"]["best_hypothesis"] == case["best_hypothesis"]
            assert r["summary"]["n_iterations"] == case["n_iterations"]
            assert r["summary"]["n_models"] == case["n_models"]
            assert np.packbits(r["mask"]).tobytes().hex() == case["mask_hex"]
            Tg = np.array([float.fromhex(x) for x in case["T_colmajor_hex"]]).reshape(4, 4).T
            assert rel_frob(r["T"], Tg) <= 1e-4
            assert [float(x).hex() for x in r["T"].T.reshape(16)] == case["T_colmajor_hex"]


@pytest.mark.parametrize("N,outl,noise,seed", [(20, 0.0, 0.0, 1), (64, 0.1, 0.3, 2), (100, 0.3, 0.5, 3), (512, 0.3, 0.5, 4242),
                                                (777, 0.5, 1.0, 5), (3000, 0.2, 0.5, 6),
                                                (4500, 0.2, 0.5, 7)])   # > 4096 points: more than one 64-word group of the inlier mask
def test_random_scenes_both_modes(N, outl, noise, seed):
    X, uv, T, inl = M.make_scene(N=N, outlier_frac=outl, noise_px=noise, seed=seed)
    with capi.Chip(256) as chip:
        check_against_oracle(chip, X, uv, seed=seed)                              # reference-faithful adaptive mode
        check_against_oracle(chip, X, uv, seed=seed + 100, n_hypotheses=200)      # benchmark mode
        check_against_oracle(chip, X, uv, seed=seed, use_mle=0, n_hypotheses=64)  # inlier-count quality measure


@pytest.mark.parametrize("thresh,ratio", [(0.02, 0.90), (0.02, 0.7)])
def test_the_reference_s_other_parameter_sets(thresh, ratio):
    """The parameter sets the reference runs DlsPnpWithRansac with besides production's 0.03 / 0.7 (DlsPnpWithRansac.cpp:207-212):
    its manual test's 0.02 / 0.90 (unittest_theia.cpp:489-494: at most ComputeMaxIterations(15, .9) = 20 iterations) and 0.02 / 0.7
    (unittest_theia.cpp:1283-1287, Cerebro.cpp:1976-1980).  Adaptive mode, both samplers, clean and cluttered scenes; also batched."""
    with capi.Chip(256) as chip:
        for N, outl, noise, seed in [(300, 0.04, 0.3, 21), (512, 0.3, 0.5, 4242), (64, 0.0, 0.0, 3), (2000, 0.08, 0.4, 8)]:
            X, uv, T, inl = M.make_scene(N=N, outlier_frac=outl, noise_px=noise, seed=seed)
            for sampler in (capi.CHIP_SAMPLER_FRESH, capi.CHIP_SAMPLER_THEIA_PERSISTENT):
                g, o = check_against_oracle(chip, X, uv, seed=seed, error_thresh=thresh, min_inlier_ratio=ratio, sampler=sampler)
                assert 5 <= g["summary"]["n_iterations"] <= (20 if ratio == 0.90 else 50)
        Xa, uva, _, _ = M.make_scene(N=300, outlier_frac=0.04, noise_px=0.3, seed=21)
        Xb, uvb, _, _ = M.make_scene(N=150, outlier_frac=0.1, noise_px=0.3, seed=22)
        p = gparams(error_thresh=thresh, min_inlier_ratio=ratio)
        rs = chip.pnp_ransac_batch([(Xa, uva), (Xb, uvb)], p, seeds=[21, 22])
        for (X, uv), sd, r in zip([(Xa, uva), (Xb, uvb)], [21, 22], rs):
            o = O.pnp_ransac(X, uv, O.ransac_params(seed=sd, error_thresh=thresh, min_inlier_ratio=ratio))
            assert r["summary"]["n_iterations"] == o["summary"]["n_iterations"] and r["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"]
            assert np.array_equal(r["mask"], o["mask"]) and np.array_equal(r["T"].view(np.uint64), o["T"].view(np.uint64))


def test_config3_512_correspondences_1000_hypotheses():
    """BASELINE config 3: 512-correspondence DlsPnpWithRansac, 1k hypotheses."""
    X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    with capi.Chip(256) as chip:
        g, o = check_against_oracle(chip, X, uv, seed=4242, n_hypotheses=1000)
        assert g["summary"]["n_models"] > 100
        assert rel_frob(g["T"][:3, :3], T[:3, :3]) < 0.02                          # and it is the right pose
        assert (g["mask"].astype(bool) & inl).sum() > 0.9 * inl.sum()


def test_role_swap_like_the_caller():
    """Cerebro.cpp:1518,1572: PNP(a->b) and PNP(b->a) on the same match set; op2 is inverted by the caller (:1582)."""
    X, uv, T, inl = M.make_scene(N=400, outlier_frac=0.1, noise_px=0.3, seed=9)
    Xb = (X @ T[:3, :3].T + T[:3, 3]).astype(np.float32).astype(np.float64)
    uva = X[:, :2] / X[:, 2:3]
    with capi.Chip(256) as chip:
        g1, _ = check_against_oracle(chip, X, uv, seed=1, n_hypotheses=100)
        g2, _ = check_against_oracle(chip, Xb, uva, seed=2, n_hypotheses=100)
        assert rel_frob(np.linalg.inv(g2["T"])[:3, :3], g1["T"][:3, :3]) < 0.05


def test_edge_cases():
    X, uv, T, inl = M.make_scene(N=64, outlier_frac=0.0, noise_px=0.0, seed=1)
    with capi.Chip(256) as chip:
        r = chip.pnp_ransac(X[:19], uv[:19])
        assert r["status"] == capi.CHIP_ERR_TOO_FEW_POINTS and r["confidence"] == -1.0   # DlsPnpWithRansac.cpp:136-139
        check_against_oracle(chip, X[:20], uv[:20], seed=3)
        rng = np.random.default_rng(0)
        Xg = rng.uniform(-1, 1, (40, 3)); Xg[:, 2] = -np.abs(Xg[:, 2]) - 1.0
        check_against_oracle(chip, Xg, rng.uniform(-1, 1, (40, 2)), n_hypotheses=20)
        # degenerate: all points identical -> singular systems everywhere, must not hang or crash
        Xd = np.tile(X[:1], (30, 1)); uvd = np.tile(uv[:1], (30, 1))
        g = chip.pnp_ransac(Xd, uvd, gparams(n_hypotheses=16))
        o = O.pnp_ransac(Xd, uvd, O.ransac_params(n_hypotheses=16))
        assert g["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"]
        p = gparams(sample_size=17)
        st = chip.lib.chip_pnp_ransac(chip.h, X.ctypes.data, uv.ctypes.data, 64, p, np.empty(16).ctypes.data,
                                      capi.C.byref(capi.C.c_float()), None, None)
        assert st == capi.CHIP_ERR_UNSUPPORTED


def same_result(a, b):
    assert a["summary"] == b["summary"] or (np.isnan(a["summary"]["best_cost"]) and np.isnan(b["summary"]["best_cost"]))
    assert np.array_equal(a["mask"], b["mask"]) and a["confidence"] == b["confidence"]
    assert np.array_equal(a["T"].view(np.uint64), b["T"].view(np.uint64))


@pytest.mark.parametrize("H", [0, 300])
def test_batch_equals_individual_calls_and_oracle(H):
    """chip_pnp_ransac_batch: ragged problem sizes, own seeds, more problems than one launch holds (8): entry i must be
    bit-identical to the single call with seeds[i] (and hence to the oracle)."""
    scenes = [M.make_scene(N=n, outlier_frac=o, noise_px=0.4, seed=s)[:2]
              for n, o, s in [(64, 0.1, 1), (512, 0.3, 2), (20, 0.0, 3), (777, 0.5, 4), (130, 0.2, 5), (512, 0.3, 6), (65, 0.0, 7),
                              (200, 0.6, 8), (333, 0.1, 9), (40, 0.0, 10), (512, 0.2, 11)]]
    seeds = [1000 + 7 * i for i in range(len(scenes))]
    with capi.Chip(256) as chip:
        got = chip.pnp_ransac_batch(scenes, gparams(n_hypotheses=H), seeds=seeds)
        for (X, uv), sd, g in zip(scenes, seeds, got):
            same_result(g, chip.pnp_ransac(X, uv, gparams(n_hypotheses=H, seed=sd)))
        for i in (1, 3, 10):
            o = O.pnp_ransac(*scenes[i], O.ransac_params(n_hypotheses=H, seed=seeds[i]))
            assert np.array_equal(got[i]["mask"], o["mask"]) and np.array_equal(got[i]["T"].view(np.uint64), o["T"].view(np.uint64))
        # shared seed (seeds = NULL) and the empty batch
        g2 = chip.pnp_ransac_batch(scenes[:2], gparams(n_hypotheses=50, seed=77))
        same_result(g2[1], chip.pnp_ransac(*scenes[1], gparams(n_hypotheses=50, seed=77)))
        assert chip.pnp_ransac_batch([], gparams()) == []
        with pytest.raises(Exception):
            chip.pnp_ransac_batch([scenes[0], (scenes[0][0][:19], scenes[0][1][:19])], gparams())    # a problem with < 20 points


def test_back_substitution_forms_agree():
    """pnp_eig_score solves the real eigenvectors with straight-line code over compile-time row indices and keeps hqr2's loop form for
    the two cases that code leaves out (a non-finite entry of H / V, the overflow rescaling).  No scene reaches those, so the loop form is
    forced by its test knob in a process of its own and must pass the same fuzz (poses, masks, iteration counts bit for bit vs the oracle)."""
    import os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    from conftest import HOOKS_ENV          # the knob exists in the TEST build of the library only (cerebro_amd/lib/hooks/)
    env = dict(os.environ, CHIP_PNP_BACKSUB="loop", **HOOKS_ENV)
    r = subprocess.run([sys.executable, str(root / "scripts" / "gpu_pnp_fuzz.py"), "36"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert "TEST KNOB ACTIVE: CHIP_PNP_BACKSUB=loop" in r.stderr, r.stderr[-400:]
    assert "fuzz: 0 mismatches" in r.stdout, (r.stdout[-400:], r.stderr[-400:])


@pytest.mark.parametrize("N,outl,noise,seed", [(20, 0.0, 0.0, 1), (100, 0.3, 0.5, 3), (512, 0.3, 0.5, 4242), (3000, 0.2, 0.5, 6)])
def test_theia_persistent_sampler_mode(N, outl, noise, seed):
    """CHIP_SAMPLER_THEIA_PERSISTENT (round 5): theia::RandomSampler as written -- one permutation per estimation, carried from
    hypothesis to hypothesis (one theia::Ransac per PNP call, DlsPnpWithRansac.cpp:216-221).  The host sequences the swaps, the
    kernels read the table: same parity bar as the default mode, in every RANSAC mode and through the batched entry point."""
    X, uv, T, inl = M.make_scene(N=N, outlier_frac=outl, noise_px=noise, seed=seed)
    with capi.Chip(256) as chip:
        g1, o1 = check_against_oracle(chip, X, uv, seed=seed, sampler=capi.CHIP_SAMPLER_THEIA_PERSISTENT)
        check_against_oracle(chip, X, uv, seed=seed + 100, n_hypotheses=200, sampler=1)
        check_against_oracle(chip, X, uv, seed=seed, use_mle=0, n_hypotheses=64, sampler=1)
        g0, o0 = check_against_oracle(chip, X, uv, seed=seed + 100, n_hypotheses=200)          # the default mode is untouched by a table left behind
        if N >= 100:    # different samples from hypothesis 1 on: the two modes are different estimations of the same scene
            a = O.pnp_ransac(X, uv, O.ransac_params(seed=seed + 100, n_hypotheses=200, sampler=1))
            assert a["summary"]["best_hypothesis"] != o0["summary"]["best_hypothesis"] or not np.array_equal(a["T"], o0["T"])
        # batched: two problems of different size, own seeds, own permutations
        X2, uv2, _, _ = M.make_scene(N=max(20, N // 2), outlier_frac=outl, noise_px=noise, seed=seed + 1)
        p = gparams(n_hypotheses=96, sampler=1)
        rs = chip.pnp_ransac_batch([(X, uv), (X2, uv2)], p, seeds=[seed, seed + 7])
        for (Xi, uvi), sd, r in zip(((X, uv), (X2, uv2)), (seed, seed + 7), rs):
            o = O.pnp_ransac(Xi, uvi, O.ransac_params(n_hypotheses=96, seed=sd, sampler=1))
            assert r["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"] and np.array_equal(r["mask"], o["mask"])
            if o["summary"]["best_hypothesis"] >= 0:
                assert np.array_equal(r["T"].view(np.uint64), o["T"].view(np.uint64))
        p.sampler = 7
        with pytest.raises(capi.ChipError) as ei:
            chip.pnp_ransac(X, uv, p)
        assert ei.value.status == capi.CHIP_ERR_UNSUPPORTED
