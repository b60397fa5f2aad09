"""The oracle's fixed-tree summation order vs the order the reference's own build uses.

The reference computes `u = v.transpose() * M.leftCols(k)` with Eigen (src/Cerebro.cpp:1026-1028); Eigen is neither vendored nor
installed here, so `oracle/dot_scan.c: orc_dot_eigen_gemv_f64 / orc_ref_scan_f64_eigen_order` restate the published algorithm of
Eigen 3.3.x's row-major GEMV (one SSE2 packet accumulator of 2 doubles per output, no FMA, predux, scalar tail) for the reference's
x86-64 Release build.  The products of float32-representable values are exact in fp64, so the two orders can differ only by
summation rounding.  These tests turn "argmax identical to Eigen except for ties below 1e-15" from an argument into a check:
  (1) the emulation is what it says it is (an independent exact-arithmetic Python restatement, every packet width);
  (2) on every committed fixture and on 10^4 random + planted + duplicated rows, the Eigen-order path takes the SAME decisions
      (argmax of all three queries, accept / reject, reported index) as the oracle the GPU is bit-exact against;
  (3) the score deviation stays below 1e-15 * sqrt(D) (absolute; scores are O(1) cosines).
CPU only.
"""
import json
from fractions import Fraction
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import scenarios

GOLD = Path(__file__).resolve().parent / "golden"


def _round_add(a: float, b: float) -> float:
    return a + b                                   # IEEE double addition: one rounding


def _dot_eigen_py(v, col, packet, fma, aligned_start=0):
    """Independent restatement: python floats are IEEE doubles; an fma is emulated exactly through Fractions."""
    D = len(v)

    def madd(a, b, c):
        if fma:
            return float(Fraction(a) * Fraction(b) + Fraction(c))     # one rounding (Fraction -> float rounds to nearest even)
        return _round_add(a * b, c)

    tmp = 0.0
    j = 0
    for j in range(min(aligned_start, D)):
        tmp = _round_add(tmp, col[j] * v[j])
    j = min(aligned_start, D)
    P = packet if packet in (2, 4) else 1
    if P > 1:
        aligned_size = aligned_start + ((D - aligned_start) & ~(P - 1))
        if aligned_size > aligned_start:
            acc = [0.0] * P
            while j < aligned_size:
                for c in range(P):
                    acc[c] = madd(col[j + c], v[j + c], acc[c])
                j += P
            red = acc[0] + acc[1] if P == 2 else (acc[0] + acc[1]) + (acc[2] + acc[3])
            tmp = _round_add(tmp, red)
    while j < D:
        tmp = _round_add(tmp, col[j] * v[j])
        j += 1
    return 0.0 + 1.0 * tmp


@pytest.mark.parametrize("D", [1, 2, 3, 7, 64, 255, 1024])
@pytest.mark.parametrize("packet,fma", [(2, False), (4, True), (4, False), (1, False)])
def test_emulation_is_the_stated_order(D, packet, fma):
    rng = np.random.default_rng(D * 7 + packet)
    v = rng.standard_normal(D).astype(np.float32).astype(np.float64)
    col = rng.standard_normal(D).astype(np.float32).astype(np.float64)
    for start in (0, 1):
        got = oracle_lib.dot_eigen_gemv(v, col, packet, fma, start)
        want = _dot_eigen_py(list(map(float, v)), list(map(float, col)), packet, fma, start)
        assert got.hex() == float(want).hex()
    if packet == 1:      # packet 1 is the plain sequential chain the cpu_baseline port uses (orc_dot_seq_f64)
        import ctypes as C
        lib = oracle_lib.load()
        lib.orc_dot_seq_f64.restype = C.c_double
        seq = float(lib.orc_dot_seq_f64(oracle_lib._p(v), oracle_lib._p(col), C.c_int32(D)))
        assert oracle_lib.dot_eigen_gemv(v, col, 1, False).hex() == seq.hex()


def _decisions_tree(db, l):
    orc = oracle_lib.LoopOracle(db)
    return orc.tick(l)


def _decisions_eigen(db64, l, packet=2, fma=False, params=None):
    p = params or oracle_lib.default_params()
    k = l - p.lag
    maxv, arg, (u, um, umm) = oracle_lib.ref_scan_f64_eigen_order(db64, k, db64[l - 1], db64[l - 2], db64[l - 3], packet, fma, nthreads=4)
    found = int(abs(arg[0] - arg[1]) < p.locality and abs(arg[0] - arg[2]) < p.locality and maxv[0] > p.thresh)   # Cerebro.cpp:1056
    return dict(argmax=[int(x) for x in arg], maxv=[float(x) for x in maxv], found=found,
                idx_prev=int(arg[0]) if found else -1), (u, um, umm)


def _compare(db, ls, stats):
    db64 = db.astype(np.float64)
    D = db.shape[1]
    for l in ls:
        t = _decisions_tree(db, l)
        if t["status"] != 2:
            continue
        for packet, fma in ((2, False), (4, True)):            # the reference's SSE2 build, and an AVX2+FMA build of the same code
            e, (u, um, umm) = _decisions_eigen(db64, l, packet, fma)
            assert e["argmax"] == t["argmax"], (l, packet, e, t)
            assert e["found"] == t["found"] and e["idx_prev"] == t["idx_prev"], (l, packet, e, t)
            dev = max(abs(a - b) for a, b in zip(e["maxv"], t["maxv"]))
            stats["max_dev"] = max(stats["max_dev"], dev)
            assert dev <= 1e-15 * np.sqrt(D), (l, dev)
            if packet == 2:
                # every column, not only the winners
                full = oracle_lib.scores(db, l - 50, db[l - 1], nthreads=4)
                stats["max_dev_all"] = max(stats["max_dev_all"], float(np.max(np.abs(full - u))))
                assert np.max(np.abs(full - u)) <= 1e-15 * np.sqrt(D)
        stats["ticks"] += 1


def test_eigen_order_takes_the_oracles_decisions_on_every_fixture():
    g = json.loads((GOLD / "dot_scan_golden.json").read_text())
    stats = dict(max_dev=0.0, max_dev_all=0.0, ticks=0)
    for case in g["cases"]:
        plants = [tuple(p) for p in case["plants"]]
        db = scenarios.build_db(case["seed"], case["N"], case["D"], plants)
        sched = case["schedule"]
        fire = {fl[0] + 1 for fl in case["found_loops"]}                 # idx_curr = l - 1
        ls = sorted(set(sched[::max(1, len(sched) // 25)]) | (fire & set(sched)))
        _compare(db, ls, stats)
    assert stats["ticks"] >= 30
    print(f"fixtures: {stats['ticks']} ticks, max |score deviation| winners {stats['max_dev']:.3e}, all columns {stats['max_dev_all']:.3e}")


def test_eigen_order_on_10k_random_planted_and_duplicated_rows():
    """4096-D x 10 053 rows (BASELINE config 2's size): random rows, planted revisits (the accept rule fires), exact duplicates (the
    last-index rule decides) -- the Eigen-order path and the oracle's tree order agree on every argmax and every decision."""
    D, N = 4096, 10_053
    plants, loops, ties = scenarios.loop_plants(N, 12, seed=2024)
    db = scenarios.build_db(20190412, N, D, plants)
    stats = dict(max_dev=0.0, max_dev_all=0.0, ticks=0)
    ls = sorted({lp[0] for lp in loops} | {N, N - 3, 5000, 2003})
    _compare(db, ls, stats)
    assert stats["ticks"] == len(ls)
    # duplicates: both orders give bit-identical scores for bit-identical rows, so the LAST duplicate wins in both
    s, t1, t2 = ties[0]
    l, q, p = loops[0]
    e, _ = _decisions_eigen(db.astype(np.float64), l)
    assert e["argmax"][0] == t2
    print(f"10k: {stats['ticks']} ticks, max |score deviation| winners {stats['max_dev']:.3e}, all columns {stats['max_dev_all']:.3e} "
          f"(bound 1e-15 * sqrt(D) = {1e-15 * np.sqrt(D):.1e})")


def test_where_the_orders_could_differ_is_a_measure_zero_tie():
    """Two DISTINCT rows whose scores differ by less than the summation noise are the only way the two orders can pick different
    indices.  Quantify the gap between the best and the second-best score of every tick above: it is > 1e-6, eleven orders of
    magnitude above the 6e-14 bound -- on descriptors like these the selection does not depend on the summation order."""
    D, N = 4096, 3000
    plants, loops, _ = scenarios.loop_plants(N, 6, seed=5, with_ties=False)
    db = scenarios.build_db(99, N, D, plants)
    gaps = []
    for l in [lp[0] for lp in loops] + [N, 1500]:
        sc, ix = oracle_lib.scan_topk(db, l - 50, db[[l - 1]], 2)
        gaps.append(sc[0][0] - sc[0][1])
    assert min(gaps) > 1e-6


@pytest.mark.parametrize("D,k", [(4096, 1203), (4095, 301), (130, 77), (2, 9)])
def test_eigen_shaped_gemv3_is_the_eigen_order_bit_for_bit(D, k):
    """bench.py's headline cpu_baseline (orc_ref_scan_f64_eigen_gemv3: three separate GEMVs, four rows at a time, one SSE2 Packet2d
    accumulator per row -- the shape Eigen 3.3's row-major kernel runs in the reference's build) is the Eigen-order emulation bit for
    bit: every score, the three maxima and the three last-index argmax, single- and multi-threaded, odd D (scalar tail) included."""
    Dp = D + (D % 4 and 4 - D % 4)
    M = oracle_lib.synth_rows(11, range(k + 3), Dp)[:, :D].astype(np.float64).copy()
    M[k // 2] = M[k // 3]                                   # an exact duplicate: the LAST index attaining the maximum must win
    v, vm, vmm = M[k + 2].copy(), M[k + 1].copy(), M[k // 3].copy()
    want = oracle_lib.ref_scan_f64_eigen_order(M, k, v, vm, vmm, 2, False)
    for nt in (1, 3):
        got = oracle_lib.ref_scan_f64_eigen_gemv3(M, k, v, vm, vmm, nt)
        for a, b in zip(want[2], got[2]):
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
        assert np.array_equal(want[0].view(np.uint64), got[0].view(np.uint64)) and np.array_equal(want[1], got[1])
    if D >= 100:                                            # (near-)unit-norm rows: the duplicate of the query row is the maximum
        assert got[1][2] == k // 2
