"""Double-row storage mode (SURVEY App. B "store that DB as fp64"): the reference's M is MatrixXd (src/Cerebro.cpp:946) and
the only 4096-D model, ReljaNetVLAD, emits genuine float64 (numpy matmul with the WPCA matrix,
scripts/whole_image_desc_compute_server.py:148-149).  Bar: indices and scores bit-exact vs the f64 oracle
(orc_dot_tree_f64: fma chains in the fixed lane order + butterfly)."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_tick(g, o):
    g = g.as_dict() if hasattr(g, "as_dict") else g
    for key in ("status", "found", "idx_curr", "idx_prev", "argmax"):
        assert g[key] == o[key], (key, g, o)
    assert float(g["score"]).hex() == float(o["score"]).hex()
    assert [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]


def relja_like(seed, N, D, plants=()):
    """float64 descriptors that are NOT float32-representable: unit-norm rows of a float64 matmul, like
    np.matmul(u, WPCA_M) + WPCA_b followed by /= norm (server.py:148-149).  Planted rows are noisy copies / duplicates."""
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((64, D))
    db = rng.standard_normal((N, 64)) @ W + 0.01 * rng.standard_normal(D)
    for dst, src, kind in sorted(plants):
        db[dst] = db[src] if kind == 2 else db[src] + 0.2 * np.linalg.norm(db[src]) / np.sqrt(D) * rng.standard_normal(D)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    assert not np.array_equal(db.astype(np.float32).astype(np.float64), db)
    return db


@pytest.mark.parametrize("D,N", [(4, 300), (250 * 4, 700), (1024, 900), (4096, 1200), (6824, 400), (8192, 400), (8200, 200), (10240, 300)])
def test_f64_topk_parity_shapes(D, N):
    plants, loops, ties = scenarios.loop_plants(N, 3, seed=D)
    db = relja_like(D, N, D, plants)
    with capi.Chip(D, storage="f64") as chip:
        assert chip.info()["storage_bytes"] == 8
        chip.append_f64(db)
        rows = [N - 1, N - 2, N - 3, loops[0][1]]
        # 3 x 8192 x 8 B = 192 KiB of queries do not fit the 160 KiB of LDS: the queries beyond what fits are read in place
        # (db_scan_topk_wide) -- D = 8192 is the reference's default descriptor size (src/Cerebro.cpp:1021)
        for nq in (1, 2, 3, 4):
            for K in (1, 8, 16):
                for k in (0, 1, 7, N - 50, N):
                    want = oracle_lib.scan_topk_f64(db, k, db[rows[:nq]], K)
                    got = chip.query_rows(k, rows[:nq], K)
                    assert np.array_equal(got[1], want[1]), (nq, K, k)
                    assert np.array_equal(bits(got[0]), bits(want[0])), (nq, K, k)
        q = relja_like(5, 3, D)
        got, want = chip.query_vectors_f64(N, q, 8), oracle_lib.scan_topk_f64(db, N, q, 8)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        assert chip.read_rows_f64([0, N // 2, N - 1]).tobytes() == db[[0, N // 2, N - 1]].tobytes()
        u = chip.query_scores(N - 50, N - 1)
        assert np.array_equal(bits(u), bits(oracle_lib.scores(db, N - 50, db[N - 1])))
        with pytest.raises(capi.ChipError) as e:
            chip.read_rows([0])                      # double rows do not fit a float buffer
        assert e.value.status == capi.CHIP_ERR_NOT_F32
        with pytest.raises(capi.ChipError) as e:
            chip.query_batch(N, np.zeros((4, D), np.float32), 4)    # the MFMA many-query mode is float-only
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED


def test_f64_tick_sequence_and_auto_switch():
    """chip_create (no flag): the first append of a non-float32 descriptor into the EMPTY DB turns it into a double-row DB
    -- no rounding, no error; INTEGRATION.md 2 no longer needs CHIP_APPEND_ALLOW_ROUNDING for ReljaNetVLAD."""
    D, N = 1024, 1500
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=42)
    db = relja_like(7, N, D, plants)
    assert ties
    orc = oracle_lib.LoopOracle64(db)
    with capi.Chip(D) as chip:
        assert chip.info()["storage_bytes"] == 4
        assert chip.append_f64(db[:1]) == 0
        assert chip.info()["storage_bytes"] == 8 and chip.info()["lossy_rows"] == 0
        chip.append_f64(db[1:700])
        chip.append_f32(db[700:701].astype(np.float32))           # float rows widen exactly into a double DB
        dbm = db.copy()
        dbm[700] = db[700].astype(np.float32)
        chip.append_f64(db[701:])
        orc = oracle_lib.LoopOracle64(dbm)
        n_found = 0
        sched = [1, 3, 5, 30, 55, 57, 58, 61] + list(range(64, N + 1, 3))
        ls = set(sched)
        for l, _, _ in loops:
            ls -= {l - 1, l - 2}
            ls.add(l)
        for l in sorted(ls):
            o = orc.tick(l)
            same_tick(chip.loop_tick(l), o)
            n_found += o["found"]
        assert n_found >= len(loops)
        s, t1, t2 = ties[0]
        l, q, p = loops[0]
        sc, ix = chip.query_rows(l - 50, [q], 3)
        assert list(ix[0]) == [t2, t1, s] and sc[0][0] == sc[0][1] == sc[0][2]
    # a float32-valued stream stays a float DB; a later genuinely-double row is then an error (not silently rounded)
    f32db = scenarios.build_db(3, 64, D, [])
    with capi.Chip(D) as chip:
        chip.append_f64(f32db.astype(np.float64))
        assert chip.info()["storage_bytes"] == 4
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(db[:1])
        assert e.value.status == capi.CHIP_ERR_NOT_F32 and chip.size() == 64
    # CHIP_CREATE_STORE_F32 never switches
    with capi.Chip(D, storage="f32") as chip:
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(db[:1])
        assert e.value.status == capi.CHIP_ERR_NOT_F32 and chip.size() == 0 and chip.info()["storage_bytes"] == 4
    # D too large for two double queries in LDS (D > 10 240, the float limit as well): explicit request refused
    with pytest.raises(capi.ChipError) as e:
        capi.Chip(10244, storage="f64")
    assert e.value.status == capi.CHIP_ERR_UNSUPPORTED


def test_f64_8192d_tick_sequence():
    """The reference's default descriptor size (8192, src/Cerebro.cpp:1021) as genuine float64 rows: the automatic switch to double rows
    works there too and every tick is bit-exact vs the f64 oracle (two queries staged in LDS, the third read in place)."""
    D, N = 8192, 900
    plants, loops, ties = scenarios.loop_plants(N, 4, seed=8192)
    db = relja_like(11, N, D, plants)
    orc = oracle_lib.LoopOracle64(db)
    with capi.Chip(D) as chip:
        chip.append_f64(db[:100])
        assert chip.info()["storage_bytes"] == 8 and chip.info()["lossy_rows"] == 0
        chip.append_f64(db[100:])
        ls = set([1, 30, 55, 57, 58, 61] + list(range(64, N + 1, 3)))
        for l, _, _ in loops:
            ls -= {l - 1, l - 2}
            ls.add(l)
        n_found = 0
        for l in sorted(ls):
            o = orc.tick(l)
            same_tick(chip.loop_tick(l), o)
            n_found += o["found"]
        assert n_found >= len(loops)


def test_f64_4096d_100k_full_oracle_parity():
    """BASELINE config 3's DB size with ReljaNetVLAD-like float64 descriptors: append_f64 succeeds unrounded and the tick is
    bit-exact vs the f64 oracle over the whole prefix."""
    D, N = 4096, 100_053
    l = N
    q, p = l - 1, 41_234
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 5, p, 2)]
    rng = np.random.default_rng(20190412)
    W = rng.standard_normal((64, D))
    db = np.empty((N, D), dtype=np.float64)
    for a in range(0, N, 8192):
        b = min(N, a + 8192)
        db[a:b] = rng.standard_normal((b - a, 64)) @ W
    for dst, src, kind in sorted(plants):
        db[dst] = db[src] if kind == 2 else db[src] + 0.2 * np.linalg.norm(db[src]) / np.sqrt(D) * rng.standard_normal(D)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    with capi.Chip(D, capacity_hint=N) as chip:
        for a in range(0, N, 20_000):
            chip.append_f64(db[a:a + 20_000])
        assert chip.info()["storage_bytes"] == 8 and chip.info()["lossy_rows"] == 0 and chip.size() == N
        r = chip.loop_tick(l)
        wsc, wix = oracle_lib.scan_topk_f64(db, l - 50, db[[l - 1, l - 2, l - 3]], 8, nthreads=min(os.cpu_count() or 1, 64))
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1 and r.idx_prev == p + 5 and r.idx_curr == q
        assert list(r.argmax) == list(wix[:, 0]) and [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        got = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(got[1], wix) and np.array_equal(bits(got[0]), bits(wsc))


@pytest.mark.parametrize("G", [2, 8])
def test_f64_group(G):
    D, N = 1024, 1300
    plants, loops, _ = scenarios.loop_plants(N, 4, seed=77)
    db = relja_like(9, N, D, plants)
    orc = oracle_lib.LoopOracle64(db)
    with capi.Chip(D, devices=[0] * G) as chip:
        chip.append_f64(db[:500])              # every device takes the same storage decision
        chip.append_f64(db[500:])
        assert chip.info()["storage_bytes"] == 8
        for l in scenarios.default_schedule(N):
            same_tick(chip.loop_tick(l), orc.tick(l))
        assert chip.read_rows_f64([3, 500, N - 1]).tobytes() == db[[3, 500, N - 1]].tobytes()


def test_f64_argmax_vs_the_reference_arithmetic_near_ties():
    """VERDICT r5 next 8 / weak 1.  On DOUBLE rows the device (and the oracle that defines it) rounds each term ONCE (fma chains in a fixed
    lane order); the reference's SSE2 build of Eigen rounds it twice (mul, then add) in another order (oracle/dot_scan.c:78-99 vs
    orc_ref_scan_f64_eigen_gemv3).  So on genuinely float64 descriptors "bit-exact" means bit-exact vs the ORACLE; against the reference's
    bits the scores differ in the last few ulps and the argmax can differ only where best and second best are closer than that.
    Measured here: (a) natural float64 descriptors -- how close best and second best ever get, and that the two arithmetics pick the
    same row on every query; (b) adversarial families of near-duplicates whose scores sit within a few 1e-16 of each other -- how
    often the picks differ and at what true margin (80-bit reference).  The numbers go to gpurun_out/r06/f64_argmax_margin.json."""
    import json
    from pathlib import Path
    D, N, NQ = 4096, 1600, 150
    db = relja_like(11, N, D)
    qs = relja_like(12, NQ, D)
    ld = db.astype(np.longdouble)

    def picks(chip, dbm, q):
        k = dbm.shape[0]
        sc, ix = chip.query_vectors_f64(k, q[None, :], 2)                       # device: top-2 in (score desc, index desc) order
        maxv, arg, (u, _, _) = oracle_lib.ref_scan_f64_eigen_gemv3(dbm, k, q, q, q)   # the reference's arithmetic + last-index argmax
        return int(ix[0][0]), float(sc[0][0] - sc[0][1]), int(arg[0])

    with capi.Chip(D, storage="f64") as chip:
        chip.append_f64(db)
        nat_margin, nat_flips = [], 0
        for q in qs:
            g, m, e = picks(chip, db, q)
            nat_margin.append(m)
            nat_flips += g != e
        assert nat_flips == 0
        assert min(nat_margin) > 1e-9                      # natural margins are ten million times the arithmetic's reach

    rng = np.random.default_rng(5)
    flips, trials, flip_margins, fam_spread = 0, 0, [], []
    with capi.Chip(D, storage="f64") as chip:
        fam = []
        for t in range(40):                                 # 40 families of 24 near-duplicates of a row that scores ~0.9 against its query
            q = qs[t]
            base = q + 0.45 * np.linalg.norm(q) / np.sqrt(D) * rng.standard_normal(D)
            base /= np.linalg.norm(base)
            rows = np.stack([base * (1.0 + m * 2.0 ** -52) for m in rng.permutation(24)])
            fam.append((q, rows))
        dbm = np.concatenate([db[:200]] + [r for _, r in fam])
        chip.append_f64(dbm)
        ldm = dbm.astype(np.longdouble)
        for t, (q, rows) in enumerate(fam):
            g, m, e = picks(chip, dbm, q)
            lo = 200 + 24 * t
            assert lo <= g < lo + 24 and lo <= e < lo + 24                      # both pick a member of the query's own family
            true = ldm[lo:lo + 24] @ q.astype(np.longdouble)
            fam_spread.append(float(true.max() - true.min()))
            trials += 1
            if g != e:
                flips += 1
                flip_margins.append(abs(float(true[g - lo] - true[e - lo])))
    # a flip needs the two candidates to be closer than the two arithmetics differ: a few ulps of a 4096-term sum near 0.9
    assert all(m < 2e-14 for m in flip_margins), flip_margins
    rep = {"D": D, "natural": {"queries": NQ, "rows": N, "flips": nat_flips, "min_margin_best_vs_second": min(nat_margin), "median_margin": float(np.median(nat_margin))},
           "near_tie_families": {"families": trials, "members": 24, "true_score_spread_max": max(fam_spread), "flips": flips,
                                 "largest_true_margin_of_a_flip": max(flip_margins) if flip_margins else 0.0},
           "what": "device (= oracle: one fused multiply-add per term, fixed lane order) vs the reference's Eigen 3.3 SSE2 GEMV arithmetic (mul + add, "
                   "Packet2d order) on genuinely float64 descriptors; true margins from 80-bit dot products"}
    out = Path(__file__).resolve().parent.parent / "gpurun_out" / "r06"
    try:
        out.mkdir(parents=True, exist_ok=True)
        (out / "f64_argmax_margin.json").write_text(json.dumps(rep, indent=1))
    except OSError:
        pass
    print(json.dumps(rep))
