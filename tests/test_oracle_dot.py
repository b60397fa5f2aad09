"""CPU tests pinning oracle/dot_scan.c (the restatement of Cerebro.cpp:903-1103) against
(a) an independent numpy mirror, (b) exact arithmetic, (c) the reference-faithful fp64 path,
(d) the semantics listed in SURVEY.md Appendix B, (e) the committed golden fixture."""
import json
import math
from fractions import Fraction
from pathlib import Path

import numpy as np
import pytest

import np_mirror
import oracle_lib
import scenarios

GOLD = Path(__file__).parent / "golden"


def test_generator_matches_mirror(oracle):
    for seed, row, e in [(1, 0, 0), (20190412, 12345, 4095), (7, 999999, 17), (2**63 + 5, 3, 1)]:
        assert oracle.orc_synth_i32(seed, row, e) == np_mirror.synth_i32(seed, row, e)
    D = 260
    for kind, src in [(0, -1), (1, 4), (2, 4)]:
        a = oracle_lib.synth_rows(99, [17], D, [(17, src, kind)] if kind else [])[0]
        b = np_mirror.synth_row(99, 17, D, kind, src)
        assert a.tobytes() == b.tobytes()


def test_generator_statistics():
    D = 4096
    rows = oracle_lib.synth_rows(5, range(64), D)
    norms = np.linalg.norm(rows.astype(np.float64), axis=1)
    assert np.all(np.abs(norms - 1.0) < 0.05)            # ~unit L2 norm like NetVLAD output (predict_utils.py:59-61)
    g = rows.astype(np.float64) @ rows.astype(np.float64).T
    off = g[~np.eye(64, dtype=bool)]
    assert np.abs(off).max() < 0.2                        # no accidental > 0.85
    planted = oracle_lib.synth_rows(5, [100], D, [(100, 3, 1)])[0].astype(np.float64)
    cos = planted @ rows[3].astype(np.float64) / np.linalg.norm(planted) / norms[3]
    assert 0.95 < cos < 0.995                             # 5/sqrt(26) = 0.98


@pytest.mark.parametrize("D", [4, 252, 256, 260, 512, 1000, 4096])
def test_dot_tree_matches_mirror_bitexact(D):
    rng = np.random.default_rng(D)
    q = rng.standard_normal(D).astype(np.float32)
    rows = rng.standard_normal((5, D)).astype(np.float32)
    mirror = np_mirror.dot_tree(q, rows)
    for i in range(5):
        assert oracle_lib.dot_tree(q, rows[i]) == mirror[i]


def test_dot_tree_close_to_exact_and_to_sequential(oracle):
    """Products of fp32 values are exact in fp64, so only summation order differs between any two
    implementations (Eigen's included).  Check the fixed tree against the exactly rounded sum and the
    plain sequential sum: |diff| <= ~D*eps*sum|a*b| (in practice a few ulp)."""
    D = 4096
    rng = np.random.default_rng(3)
    q = (rng.standard_normal(D) / 64).astype(np.float32)
    r = (rng.standard_normal(D) / 64).astype(np.float32)
    tree = oracle_lib.dot_tree(q, r)
    exact = float(sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(q, r)))
    qd, rd = q.astype(np.float64), r.astype(np.float64)
    seq = float(oracle.orc_dot_seq_f64(qd.ctypes.data, rd.ctypes.data, D))
    bound = 64 * np.finfo(np.float64).eps * float(np.abs(qd * rd).sum())
    assert abs(tree - exact) <= bound
    assert abs(seq - exact) <= bound * 64
    assert math.fsum(qd * rd) == pytest.approx(exact, abs=1e-300, rel=1e-16)


def test_topk_order_and_padding():
    D = 256
    db = scenarios.build_db(11, 40, D, [(30, 7, 2), (35, 7, 2)])       # rows 7, 30, 35 identical
    q = oracle_lib.synth_rows(11, [1000], D, [(1000, 7, 1)])          # noisy copy of row 7
    sc, ix = oracle_lib.scan_topk(db, 40, q, 5)
    assert list(ix[0][:3]) == [35, 30, 7]                              # ties -> largest index first
    assert sc[0][0] == sc[0][1] == sc[0][2]
    u = np_mirror.dot_tree(q[0], db)
    msc, mix = np_mirror.topk(u, 5)
    assert np.array_equal(mix, ix[0]) and np.array_equal(msc, sc[0])
    sc, ix = oracle_lib.scan_topk(db, 3, q, 5)                         # k < K -> padded
    assert list(ix[0][3:]) == [-1, -1] and np.all(np.isneginf(sc[0][3:]))
    sc, ix = oracle_lib.scan_topk(db, 0, q, 2)
    assert list(ix[0]) == [-1, -1]


def test_unit_generator_is_unit_l2_and_matches_its_definition(oracle):
    """SURVEY.md 8d's data: unit-L2 rows.  The definition restated in numpy (exact integer sum of squares, two correctly rounded fp64
    operations, one multiply, one conversion) gives the oracle's bits; norms are 1 to fp32 rounding; planted rows keep their angles."""
    D, seed = 4096, 20190412
    plants = [(100, 3, 1), (101, 3, 2)]
    rows = [0, 3, 17, 100, 101, 999_999]
    got = oracle_lib.synth_rows(seed, rows, D, plants, unit=True)
    pm = {d: (s, k) for d, s, k in plants}
    for out, r in zip(got, rows):
        src, kind = pm.get(r, (-1, 0))
        own = np.array([oracle.orc_synth_i32(seed, r, e) for e in range(D)], dtype=np.int64)
        if kind:
            sv = np.array([oracle.orc_synth_i32(seed, src, e) for e in range(D)], dtype=np.int64)
            v = sv if kind == 2 else 5 * sv + own
        else:
            v = own
        S = int((v * v).sum())
        assert S < 2**53
        inv = 1.0 / math.sqrt(float(S))
        want = (v.astype(np.float64) * inv).astype(np.float32)
        assert out.tobytes() == want.tobytes()
    norms = np.linalg.norm(got.astype(np.float64), axis=1)
    assert np.all(np.abs(norms - 1.0) < 2e-7)            # unit vectors to fp32 rounding (the plain generator: ~1 % rms spread)
    g = got.astype(np.float64) @ got.astype(np.float64).T
    assert 0.975 < g[3, 1] < 0.985                       # row 100 = unit(5 x3 + x100): cos = 5 / sqrt(26) = 0.9806
    assert got[4].tobytes() == got[1].tobytes()          # kind 2: the unit vector of src, bit for bit
    assert abs(g[0, 2]) < 0.1                            # unrelated rows: N(0, 1/64)
    # the direction is the plain generator's: same integers, another scale
    plain = oracle_lib.synth_rows(seed, [17], D)[0].astype(np.float64)
    assert abs(plain @ got[2].astype(np.float64) / np.linalg.norm(plain) - 1.0) < 1e-6


def test_scan_synth_unit_matches_materialised_and_threads():
    D, N = 128, 700
    plants = [(300, 20, 1), (301, 21, 2), (650, 20, 2)]
    db = oracle_lib.synth_rows(21, range(N), D, plants, unit=True)
    q = db[[N - 1, N - 2, 300]]
    a = oracle_lib.scan_topk(db, N - 50, q, 8)
    for nt in (1, 3, 8):
        b = oracle_lib.scan_topk_synth(21, N - 50, D, q, 8, plants, nthreads=nt, unit=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[1][2][0] == 300 and abs(a[0][2][0] - 1.0) < 1e-6      # row 300 against itself: a unit vector
    assert a[1][2][1] == 20 and 0.95 < a[0][2][1] < 0.995          # ... then row 20, which it is a noisy copy of (row 650 is outside the prefix)


def test_scan_synth_matches_materialised_and_threads():
    D, N = 128, 700
    plants = [(300, 20, 1), (301, 21, 2), (650, 20, 2)]
    db = scenarios.build_db(21, N, D, plants)
    q = db[[N - 1, N - 2, 300]]
    a = oracle_lib.scan_topk(db, N - 50, q, 8)
    for nt in (1, 3, 8):
        b = oracle_lib.scan_topk_synth(21, N - 50, D, q, 8, plants, nthreads=nt)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_tick_semantics_appendix_b():
    D, N = 64, 200
    plants, loops, _ = scenarios.loop_plants(N, 1, seed=4, with_ties=False)
    db = scenarios.build_db(8, N, D, plants)
    orc = oracle_lib.LoopOracle(db)
    # (1) fewer than 3 new rows: skipped and last_l NOT advanced (Cerebro.cpp:962-966)
    assert orc.tick(2)["status"] == 0 and orc.state.last_l == 0
    # (4) first productive tick needs l >= 56 (k = l-50 > 5)
    assert orc.tick(55)["status"] == 1 and orc.state.last_l == 55
    assert orc.tick(57)["status"] == 0 and orc.state.last_l == 55   # only 2 new rows
    r = orc.tick(58)
    assert r["status"] == 2 and orc.state.last_l == 58
    # (3) the 50 newest rows are never searched: argmax < k
    assert max(r["argmax"]) < 58 - 50
    # (2) a jump of > 3 rows: only rows l-1,l-2,l-3 are queries
    l, q, p = loops[0]
    orc2 = oracle_lib.LoopOracle(db)
    r = orc2.tick(l)
    assert r["found"] == 1 and r["idx_curr"] == q and r["idx_prev"] == p and r["argmax"] == [p, p - 1, p - 2]
    # mirror agrees on the whole record
    st = {"last_l": 0}
    m = np_mirror.loop_tick(st, db, l)
    assert m == r


def test_threshold_is_float_rounded_and_strict():
    p = oracle_lib.default_params()
    assert p.thresh == float(np.float32(0.85)) == 0.85000002384185791015625   # Cerebro.cpp:913 float vs double at :1056
    assert (p.locality, p.lag, p.min_new, p.min_k) == (12, 50, 3, 5)
    # strict '>' : a best score exactly equal to the threshold does not fire
    D = 8
    db = np.zeros((60, D), dtype=np.float32)
    db[:, 0] = 1.0
    db[0:3, 0] = np.float32(0.85)       # rows 0..2 score exactly (double)(float)0.85 against unit queries
    db[3:, 1] = 1.0; db[3:, 0] = 0.0     # everything else orthogonal ...
    db[57:60] = 0.0; db[57:60, 0] = 1.0  # ... except the three query rows
    orc = oracle_lib.LoopOracle(db)
    r = orc.tick(60)
    assert r["status"] == 2 and r["maxv"][0] == p.thresh and r["found"] == 0
    db[0:3, 0] = np.nextafter(np.float32(0.85), np.float32(1))
    assert oracle_lib.LoopOracle(db).tick(60)["found"] == 1


def test_locality_rule():
    D, N = 1024, 300
    # query rows l-1, l-2, l-3 point at rows p, p-11, p+11 -> |d| = 11 < 12 fires; 12 does not
    for delta, expect in [(11, 1), (12, 0)]:
        l = 290
        p = 100
        plants = [(l - 1, p, 1), (l - 2, p - delta, 1), (l - 3, p + delta, 1)]
        db = scenarios.build_db(3, N, D, plants)
        r = oracle_lib.LoopOracle(db).tick(l)
        assert r["argmax"] == [p, p - delta, p + delta]
        assert r["found"] == expect


def test_reference_faithful_fp64_path_agrees_with_tree():
    """orc_ref_scan_f64_colmajor is the literal Cerebro.cpp:1026-1043 (fp64 M, 3 GEMVs, maxCoeff, last-index
    loop); the tree-ordered fp32-storage scan must select the same indices, scores within summation noise."""
    D, N = 512, 600
    plants, loops, ties = scenarios.loop_plants(N, 3, seed=9)
    db = scenarios.build_db(13, N, D, plants)
    M = db.astype(np.float64)
    for l, q, p in loops:
        k = l - 50
        maxv, arg, _ = oracle_lib.ref_scan_f64_colmajor(M, k, M[l - 1], M[l - 2], M[l - 3])
        sc, ix = oracle_lib.scan_topk(db, k, db[[l - 1, l - 2, l - 3]], 1)
        assert list(arg) == list(ix[:, 0])
        assert np.allclose(maxv, sc[:, 0], rtol=0, atol=1e-13)
    if ties:
        s, t1, t2 = ties[0]
        l, q, p = loops[0]
        _, arg, _ = oracle_lib.ref_scan_f64_colmajor(M, l - 50, M[l - 1], M[l - 2], M[l - 3])
        assert arg[0] == t2                         # last index attaining the max


def test_golden_fixture(oracle):
    g = json.loads((GOLD / "dot_scan_golden.json").read_text())
    for case in g["cases"]:
        plants = [tuple(p) for p in case["plants"]]
        db = scenarios.build_db(case["seed"], case["N"], case["D"], plants)
        orc = oracle_lib.LoopOracle(db)
        got = []
        for l in case["schedule"]:
            r = orc.tick(l)
            if r["found"]:
                got.append([r["idx_curr"], r["idx_prev"], r["score"].hex()])
        assert got == case["found_loops"]
        sc, ix = oracle_lib.scan_topk(db, case["topk_k"], db[case["topk_rows"]], case["K"])
        assert ix.tolist() == case["topk_idx"]
        assert [[float(x).hex() for x in row] for row in sc] == case["topk_scores_hex"]


def test_all_cores_baseline_equals_single_thread_reference_path():
    """bench.py's cpu_baseline_all_cores (OpenMP over columns) computes the same u/um/umm, maxima and
    last-index argmax as the literal single-thread statements."""
    src = oracle_lib.synth_rows(3, range(40), 256).astype(np.float64)
    M = oracle_lib.tile_columns_omp(1003, src, 4)                   # tiled copies -> exact ties, last index must win
    assert np.array_equal(M[:40], src) and np.array_equal(M[1000], src[1000 % 40])
    v, vm, vmm = src[5].copy(), src[6].copy(), src[7].copy()
    a = oracle_lib.ref_scan_f64_colmajor_omp(M, 1003, v, vm, vmm, 4)
    b = oracle_lib.ref_scan_f64_colmajor(M, 1003, v, vm, vmm)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    assert list(a[1]) == [965, 966, 967]                             # last tiled copies of rows 5, 6, 7


# ------------------------------------------------------------------ double-row mode (orc_dot_tree_f64 / orc_loop_tick_f64)
def _fma_exact(a: float, b: float, c: float) -> float:
    """fma by exact rational arithmetic: ONE rounding of a*b + c (Fraction -> float is correctly rounded)"""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def _dot_tree_f64_py(q, row):
    D = len(q)
    acc = [0.0] * 64
    for base in range(0, D, 128):
        for L in range(64):
            for c in range(2):
                e = base + 2 * L + c
                if e < D:
                    acc[L] = _fma_exact(float(q[e]), float(row[e]), acc[L])
    m = 32
    while m >= 1:
        acc = [acc[L] + acc[L ^ m] for L in range(64)]
        m >>= 1
    return acc[0]


@pytest.mark.parametrize("D", [4, 126, 128, 130, 300, 1024])
def test_dot_tree_f64_is_the_stated_fma_chain(D):
    """The f64 definition fixes the rounding of every term (fused multiply-add) AND the order (lane L takes elements
    j*128 + 2L + c; butterfly 32..1): pinned against exact rational arithmetic, so a libm / compiler fma that double-rounds
    would be caught."""
    rng = np.random.default_rng(D)
    q = rng.standard_normal(D) / 8
    for i in range(3):
        row = rng.standard_normal(D) / 8
        got = oracle_lib.dot_tree_f64(q, row)
        assert got == _dot_tree_f64_py(q, row)
        exact = float(sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(q, row)))
        assert abs(got - exact) <= 64 * np.finfo(np.float64).eps * float(np.abs(q * row).sum())
    # the mul-then-add reading differs in the last bits for genuine doubles -- the reason the definition says fma
    diff = 0
    for i in range(20):
        row = rng.standard_normal(D) / 8
        a = oracle_lib.dot_tree_f64(q, row)
        acc = np.zeros(64)
        for base in range(0, D, 128):
            for L in range(64):
                for c in range(2):
                    e = base + 2 * L + c
                    if e < D:
                        acc[L] = acc[L] + q[e] * row[e]
        m = 32
        while m >= 1:
            acc = acc + acc[np.arange(64) ^ m]
            m >>= 1
        diff += a != acc[0]
    assert D < 100 or diff > 0


def test_f64_tick_and_topk_oracle_consistency():
    """orc_loop_tick_f64 / orc_scan_topk_f64 / orc_scores agree with each other, are thread-count independent, and on
    float32-valued data the f64 path selects what the f32 path selects (products are exact there, only the lane grouping
    differs, so scores agree to a few ulp)."""
    D, N = 1024, 600
    plants, loops, ties = scenarios.loop_plants(N, 4, seed=21)
    db32 = scenarios.build_db(22, N, D, plants)
    rng = np.random.default_rng(1)
    db = db32.astype(np.float64) * (1.0 + 2.0 ** -40 * rng.standard_normal((N, D)))      # genuine doubles
    db[[d for d, s, k in plants if k == 2]] = db[[s for d, s, k in plants if k == 2]]     # keep exact duplicates exact
    o64 = oracle_lib.LoopOracle64(db)
    found = 0
    for l in scenarios.default_schedule(N):
        r = o64.tick(l)
        if r["status"] == 2:
            k = l - 50
            sc1, ix1 = oracle_lib.scan_topk_f64(db, k, db[[l - 1, l - 2, l - 3]], 4, nthreads=1)
            sc3, ix3 = oracle_lib.scan_topk_f64(db, k, db[[l - 1, l - 2, l - 3]], 4, nthreads=3)
            assert np.array_equal(ix1, ix3) and sc1.tobytes() == sc3.tobytes()
            assert list(ix1[:, 0]) == r["argmax"] and list(sc1[:, 0]) == r["maxv"]
            u = oracle_lib.scores(db, k, db[l - 1], nthreads=2)
            assert u.max() == r["maxv"][0] and int(np.flatnonzero(u == u.max())[-1]) == r["argmax"][0]
        found += r["found"]
    assert found >= len(loops)
    a = oracle_lib.LoopOracle(db32)
    b = oracle_lib.LoopOracle64(db32.astype(np.float64))
    for l in scenarios.default_schedule(N):
        ra, rb = a.tick(l), b.tick(l)
        assert (ra["status"], ra["found"], ra["argmax"]) == (rb["status"], rb["found"], rb["argmax"])
        if ra["status"] == 2:
            assert np.allclose(ra["maxv"], rb["maxv"], rtol=1e-14, atol=1e-16)
