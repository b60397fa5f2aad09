"""GPU parity tests of the descriptor dot-product scan path: HIP kernels (through the C-ABI) vs the CPU oracle
on identical inputs.  Bar: indices AND scores bit-exact (integer/index work; fp64 sums in a fixed order)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_topk_equal(got, want):
    (gs, gi), (ws, wi) = got, want
    assert np.array_equal(gi, wi), (gi, wi)
    assert np.array_equal(bits(gs), bits(ws)), (gs, ws)


def same_tick(g, o):
    g = g.as_dict() if hasattr(g, "as_dict") else g
    for key in ("status", "found", "idx_curr", "idx_prev", "argmax"):
        assert g[key] == o[key], (key, g, o)
    assert float(g["score"]).hex() == float(o["score"]).hex()
    assert [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]


def test_golden_fixture_through_c_abi():
    g = json.loads((GOLD / "dot_scan_golden.json").read_text())
    for case in g["cases"]:
        db = scenarios.build_db(case["seed"], case["N"], case["D"], [tuple(p) for p in case["plants"]])
        with capi.Chip(case["D"]) as chip:
            assert chip.append_f64(db.astype(np.float64)) == 0       # wire type float64[] (srv:4), lossless narrowing
            got = []
            for l in case["schedule"]:
                r = chip.loop_tick(l)
                if r.found:
                    got.append([r.idx_curr, r.idx_prev, float(r.score).hex()])
            assert got == case["found_loops"]
            sc, ix = chip.query_rows(case["topk_k"], case["topk_rows"], case["K"])
            assert ix.tolist() == case["topk_idx"]
            assert [[float(x).hex() for x in row] for row in sc] == case["topk_scores_hex"]


@pytest.mark.parametrize("D,N", [(4, 300), (252, 500), (256, 700), (1000, 900), (4096, 2500), (8192, 600)])
def test_topk_parity_shapes(D, N):
    plants, loops, ties = scenarios.loop_plants(N, 3, seed=D)
    db = scenarios.build_db(1000 + D, N, D, plants)
    with capi.Chip(D) as chip:
        chip.append_f32(db)
        rows = [N - 1, N - 2, N - 3, loops[0][1]]
        for nq in (1, 2, 3, 4):
            for K in (1, 5, 8, 16):
                for k in (0, 1, 7, N - 50, N):
                    want = oracle_lib.scan_topk(db, k, db[rows[:nq]], K)
                    got = chip.query_rows(k, rows[:nq], K)
                    assert_topk_equal(got, want)
        # external query vectors take the same path
        q = oracle_lib.synth_rows(5, [10, 11, 12], D)
        assert_topk_equal(chip.query_vectors(N, q, 8), oracle_lib.scan_topk(db, N, q, 8))


@pytest.mark.parametrize("env", [{}, {"CHIP_TICK_FUSED": "0"}, {"CHIP_TICK_POLL": "0"}, {"CHIP_TICK_FUSED": "0", "CHIP_TICK_SAME_STREAM": "0"}],
                         ids=["default", "two-launch tick", "event-collected tick", "two-launch tick on the scan streams"])
def test_tick_sequence_parity_and_tie_rule(env, monkeypatch):
    """... in the product's form of the short tick (one fused launch, completion word polled) and in its documented fall-backs: the
    two-launch tick (K1 lists -> K2 merge: kernel-boundary visibility only, README `CHIP_TICK_FUSED=0`) and the event-collected one."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    D, N = 1024, 1500
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=42)
    db = scenarios.build_db(31337, N, D, plants)
    assert ties, "scenario must contain exact duplicate rows"
    orc = oracle_lib.LoopOracle(db)
    with capi.Chip(D) as chip:
        chip.append_f64(db[:700].astype(np.float64))
        chip.append_f64(db[700:].astype(np.float64))
        n_found = 0
        # irregular schedule: skips (<3 new), jumps (>3 new), first ticks too short
        sched = [1, 3, 5, 30, 55, 57, 58, 61] + list(range(64, N + 1, 3))
        extra = [l + 1 for l in sched[10::17]] + [l + 2 for l in sched[11::13]]
        ls = set(sched + [x for x in extra if x <= N])
        for l, _, _ in loops:                 # make sure every planted revisit is actually ticked
            ls -= {l - 1, l - 2}
            ls.add(l)
        for l in sorted(ls):
            o = orc.tick(l)
            g = chip.loop_tick(l)
            same_tick(g, o)
            assert chip.last_l() == orc.state.last_l
            n_found += o["found"]
        assert n_found >= len(loops)
        s, t1, t2 = ties[0]
        l, q, p = loops[0]
        sc, ix = chip.query_rows(l - 50, [q], 3)
        assert list(ix[0]) == [t2, t1, s] and sc[0][0] == sc[0][1] == sc[0][2]   # last index wins (Cerebro.cpp:1038-1043)


def test_pipelined_ticks_match_sync():
    D, N = 512, 1200
    plants, loops, _ = scenarios.loop_plants(N, 5, seed=7)
    db = scenarios.build_db(9, N, D, plants)
    sched = scenarios.default_schedule(N)
    with capi.Chip(D) as a, capi.Chip(D) as b:
        a.append_f32(db)
        b.append_f32(db)
        sync = [a.loop_tick(l).as_dict() for l in sched]
        out = []
        for base in range(0, len(sched), 32):
            chunk = sched[base:base + 32]
            for s, l in enumerate(chunk):
                b.loop_tick_enqueue(l, s)
            out += [b.loop_tick_collect(s).as_dict() for s in range(len(chunk))]
        assert out == sync
        with pytest.raises(capi.ChipError):
            b.loop_tick_collect(0)            # nothing in flight


def test_append_validation_and_errors():
    D = 256
    good = oracle_lib.synth_rows(1, range(8), D).astype(np.float64)
    with capi.Chip(D) as chip:
        assert chip.append_f64(good) == 0
        bad = good.copy()
        bad[3, 17] = 0.1                       # not representable in float32
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(bad)
        assert e.value.status == capi.CHIP_ERR_NOT_F32 and chip.size() == 8      # nothing appended
        nan = good.copy()
        nan[0, 0] = np.nan
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(nan)
        assert e.value.status == capi.CHIP_ERR_NONFINITE and chip.size() == 8
        assert chip.append_f64(bad, allow_rounding=True) == 8
        assert chip.info()["lossy_rows"] > 0
        back = chip.read_rows(range(16))
        assert back[:8].tobytes() == good.astype(np.float32).tobytes()
        assert back[8:].tobytes() == bad.astype(np.float32).tobytes()
        with pytest.raises(capi.ChipError) as e:
            chip.loop_tick(17)                 # l beyond the appended rows
        assert e.value.status == capi.CHIP_ERR_RANGE
        with pytest.raises(capi.ChipError) as e:
            chip.query_rows(5, [99], 4)
        assert e.value.status == capi.CHIP_ERR_RANGE
        with pytest.raises(capi.ChipError) as e:
            chip.query_rows(5, [1], 17)
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED


def test_device_generator_bit_identical_to_spec():
    D, N = 4096, 3000
    plants = [(100, 5, 1), (101, 6, 2), (2999, 2000, 1)]
    with capi.Chip(D) as chip:
        chip.append_synthetic(1000, 20190412, [p for p in plants if p[0] < 1000])
        chip.append_synthetic(N - 1000, 20190412, [p for p in plants if p[0] >= 1000])   # appended in two calls
        rows = [0, 1, 5, 6, 100, 101, 999, 1000, 2000, 2999]
        got = chip.read_rows(rows)
        want = oracle_lib.synth_rows(20190412, rows, D, plants)
        assert got.tobytes() == want.tobytes()


def test_growth_across_segments():
    """DB segments are 512 MiB; with D=8192 (32 KiB rows) a segment holds 16384 rows -> cross it."""
    D, N = 8192, 20000
    with capi.Chip(D, capacity_hint=100) as chip:
        chip.append_synthetic(N, 3, [(19990, 16383, 1), (19991, 16384, 1), (19992, 5, 2)])
        info = chip.info()
        assert info["rows_global"] == N and info["capacity_local"] >= N
        q = oracle_lib.synth_rows(3, [19990, 19991, 19992], D, [(19990, 16383, 1), (19991, 16384, 1), (19992, 5, 2)])
        sc, ix = chip.query_rows(N - 50, [19990, 19991, 19992], 2)
        assert list(ix[:, 0]) == [16383, 16384, 5]
        for i, r in enumerate([16383, 16384, 5]):
            row = oracle_lib.synth_rows(3, [r], D)[0]
            assert sc[i, 0] == oracle_lib.dot_tree(q[i], row)


def test_sharded_scan_matches_single():
    """Round-robin row shards (rank r keeps rows i % G == r) + merge == unsharded result, for G = 2, 3, 8.
    All shard contexts live on the one GPU of the test box; the all-gather is emulated by a device concat."""
    import torch
    D, N = 512, 1400
    plants, loops, ties = scenarios.loop_plants(N, 5, seed=11)
    db = scenarios.build_db(17, N, D, plants)
    orc_ticks = {}
    orc = oracle_lib.LoopOracle(db)
    sched = scenarios.default_schedule(N)
    for l in sched:
        orc_ticks[l] = orc.tick(l)
    for G in (2, 3, 8):
        chips = [capi.Chip(D, shard_rank=r, shard_count=G) for r in range(G)]
        try:
            K = 8
            for c in chips:
                c.append_f64(db[:333].astype(np.float64))
                c.append_f32(db[333:])
                assert c.info()["rows_local"] == len(range(c.shard_rank, N, G))
            bufs = torch.zeros((G, 3, K, 2), dtype=torch.float64, device="cuda")
            for l in sched:
                st = [c.scan_local(l, bufs[r].data_ptr(), K) for r, c in enumerate(chips)]
                assert len(set(st)) == 1
                if st[0] != capi.CHIP_TICK_SCANNED:
                    assert orc_ticks[l]["status"] == st[0]
                    continue
                for c in chips:
                    c.synchronize()
                res = [c.merge_decide(l, bufs.data_ptr(), G, K) for c in chips]
                for r in res:
                    same_tick(r, orc_ticks[l])
        finally:
            for c in chips:
                c.close()


def test_10k_full_oracle_parity():
    """BASELINE config 2 at its own size (4096-D x 10k, "dot-product + top-k only"): full CPU oracle scan vs one GPU tick and
    the top-8 lists, bit-exact.  164 MB: the DB is Infinity-Cache resident at this size."""
    D, N, seed = 4096, 10_053, 20190412
    l = N
    q, p = l - 1, 4_321
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 5, p, 2), (77, p - 1, 2)]   # a later and an earlier exact duplicate
    with capi.Chip(D, capacity_hint=N) as chip:
        chip.append_synthetic(N, seed, plants)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, plants)
        wsc, wix = oracle_lib.scan_topk_synth(seed, l - 50, D, qrows, 8, plants, nthreads=os.cpu_count() or 1)
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1 and r.idx_curr == q
        assert list(r.argmax) == list(wix[:, 0]) == [p + 5, p - 1, p - 2] and r.idx_prev == p + 5
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        for K in (1, 5, 8, 16):
            w = oracle_lib.scan_topk_synth(seed, l - 50, D, qrows, K, plants, nthreads=os.cpu_count() or 1)
            assert_topk_equal(chip.query_rows(l - 50, [l - 1, l - 2, l - 3], K), w)
        # the three score vectors u, um, umm in full (src/Cerebro.cpp:1026-1028): every column, bit for bit
        db = oracle_lib.synth_rows(seed, range(l - 50), D, plants)
        for j in range(3):
            u = chip.query_scores(l - 50, l - 1 - j)
            assert np.array_equal(bits(u), bits(oracle_lib.scores(db, l - 50, qrows[j], nthreads=os.cpu_count() or 1)))
            assert int(np.flatnonzero(u == u.max())[-1]) == r.argmax[j]      # maxCoeff + LAST index attaining it (:1035-1043)
        # ... and against the summation order of the reference's own build (Eigen 3.3 row-major GEMV, SSE2 packets, no FMA --
        # oracle/dot_scan.c orc_ref_scan_f64_eigen_order): the GPU's score vector deviates by rounding only, the selection is the same
        db64 = db.astype(np.float64)
        emaxv, earg, (eu, eum, eumm) = oracle_lib.ref_scan_f64_eigen_order(db64, l - 50, qrows[0], qrows[1], qrows[2], 2, False,
                                                                           nthreads=os.cpu_count() or 1)
        assert list(earg) == list(r.argmax)
        u0 = chip.query_scores(l - 50, l - 1)
        assert np.max(np.abs(u0 - eu)) <= 1e-15 * np.sqrt(D)
        assert abs(emaxv[0] - r.maxv[0]) <= 1e-15 * np.sqrt(D) and (emaxv[0] > capi.default_dot_params().thresh) == bool(r.found)


def test_score_vector_export_2500():
    """chip_query_scores: u = v^T M[:, :k] for one query row, checked against orc_dot_tree_f32 for EVERY column at
    4096-D x 2500 and at a ragged D."""
    for D, N in ((4096, 2500), (252, 700)):
        db = scenarios.build_db(77 + D, N, D, [])
        with capi.Chip(D) as chip:
            chip.append_f32(db)
            for k, row in ((N - 50, N - 1), (N, 0), (1, 5), (0, 5)):
                u = chip.query_scores(k, row)
                assert u.shape == (k,) and np.array_equal(bits(u), bits(oracle_lib.scores(db, k, db[row])))
            with pytest.raises(capi.ChipError) as e:
                chip.query_scores(N + 1, 0)
            assert e.value.status == capi.CHIP_ERR_RANGE


def test_100k_full_oracle_parity():
    """BASELINE config 2/3 scale (4096-D x 100k): full CPU oracle scan (threads) vs one GPU tick, bit-exact."""
    D, N, seed = 4096, 100_053, 20190412
    l = N
    q, p = l - 1, 41234
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 5, p, 2)]     # a later exact duplicate of p -> tie rule
    with capi.Chip(D, capacity_hint=N) as chip:
        chip.append_synthetic(N, seed, plants)
        chip.lib.chip_loop_reset(chip.h)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, plants)
        wsc, wix = oracle_lib.scan_topk_synth(seed, l - 50, D, qrows, 8, plants, nthreads=os.cpu_count() or 1)
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1
        assert list(r.argmax) == list(wix[:, 0]) and r.idx_prev == p + 5 and r.idx_curr == q
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        assert_topk_equal(chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8), (wsc, wix))


def test_1M_full_size_parity_and_properties():
    """BASELINE headline size (4096-D x 1M).  (a) full threaded CPU-oracle scan vs the GPU tick, bit-exact;
    (b) size-independent properties: top-k of the full prefix == merge of the top-k of two half-prefixes'
    complements (checked through prefix monotonicity), best score non-decreasing in k, sharded == unsharded."""
    D, N, seed = 4096, 1_000_053, 20190412
    l = N
    q, p = l - 1, 777_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2), (123_456, p - 1, 2)]
    ncpu = os.cpu_count() or 1
    with capi.Chip(D, capacity_hint=N) as chip:
        chip.append_synthetic(N, seed, plants)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, plants)
        assert chip.read_rows([l - 1, l - 2, l - 3]).tobytes() == qrows.tobytes()
        wsc, wix = scenarios.cached_scan_topk_synth(seed, l - 50, D, [l - 1, l - 2, l - 3], 8, plants, nthreads=min(ncpu, 128))
        assert list(r.argmax) == list(wix[:, 0])
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        # argmax = [p+4 (later duplicate of p), p-1 (123456 is an EARLIER duplicate -> loses the tie), p-2]
        assert list(r.argmax) == [p + 4, p - 1, p - 2] and r.found == 1 and r.idx_prev == p + 4
        full = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert_topk_equal(full, (wsc, wix))
        # prefix property: the top-k over [0,k1) is the top-k of the full list restricted to idx < k1 whenever
        # at least K of the full winners lie below k1; and the best score is monotone in k
        prev_best = -np.inf
        for k1 in (10, 1000, 123_457, 500_000, p - 1, p + 5, l - 50):
            sc, ix = chip.query_rows(k1, [l - 1, l - 2, l - 3], 8)
            assert np.all(ix < k1) and np.all(sc[:, 0] >= prev_best if np.isscalar(prev_best) else True)
            assert np.all(np.diff(sc, axis=1) <= 0)
            for qi in range(3):
                keep = [(s, i) for s, i in zip(full[0][qi], full[1][qi]) if i < k1]
                for (s, i), gs, gi in zip(keep, sc[qi], ix[qi]):
                    assert (s, i) == (gs, gi)
            best = sc[:, 0].copy()
            if not np.isscalar(prev_best):
                assert np.all(best >= prev_best)
            prev_best = best


def test_sharded_tick_over_rccl_world1():
    """The sharded orchestration (cerebro_amd/sharded.py) on the real RCCL backend, as far as a 1-GPU box allows:
    world_size 1, kernels + all_gather_into_tensor on torch's current stream (chip_set_stream), merge + decision."""
    import socket
    import torch
    import torch.distributed as dist
    from cerebro_amd import sharded
    torch.cuda.set_device(0)
    for attempt in range(5):     # (a port that was free a moment ago can be taken by the time the store binds it: seen once on the pool)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            break
        except (RuntimeError, OSError):
            if attempt == 4:
                raise
    try:
        D, N = 1024, 900
        plants, loops, _ = scenarios.loop_plants(N, 4, seed=3)
        db = scenarios.build_db(5, N, D, plants)
        orc = oracle_lib.LoopOracle(db)
        with capi.Chip(D, shard_rank=0, shard_count=1) as chip:
            chip.append_f32(db)
            det = sharded.ShardedLoopDetector(chip, topk=8, device=torch.device("cuda", 0))
            for l in [2, 40] + scenarios.default_schedule(N):
                o = orc.tick(l)
                g = det.tick(l)
                assert g.status == o["status"]
                if o["status"] == 2:
                    same_tick(g, o)
            # pipelined form: scans on the internal lanes overlap the all-gather + merge on torch's stream
            orc2 = oracle_lib.LoopOracle(db)
            chip.loop_reset()
            sched = scenarios.default_schedule(N)
            for base in range(0, len(sched), 20):
                chunk = sched[base:base + 20]
                sts = [det.tick_enqueue(l, s) for s, l in enumerate(chunk)]
                for s, l in enumerate(chunk):
                    o = orc2.tick(l)
                    assert sts[s] == o["status"]
                    if o["status"] == 2:
                        same_tick(det.collect(s), o)
            det.close()
    finally:
        dist.destroy_process_group()


def test_fused_tick_handoff_stress(monkeypatch):
    """The fused tick's cross-workgroup hand-off (kernels.hip fused_tick_finish: write-through stores + vmcnt(0) + ticket on the
    writers' side, an agent-scope ACQUIRE + L2-bypassing loads on the last workgroup's side) against the two-launch path
    (CHIP_TICK_FUSED=0: K1 lists -> K2 merge, ordinary kernel-boundary visibility): >= 200 000 pipelined fused ticks at full grid
    cycling over THREE query triples -- the list buffers are a ring of 64 and 64 % 3 != 0, so the launch that last used a tick's
    list buffer had a different triple and different per-workgroup bests in every workgroup: a stale entry read by the reducing
    workgroup would change argmax / maxv.  Every record is compared byte for byte with the unfused result of the same tick."""
    import ctypes
    D, seed = 4096, 77
    n_rows = 20_200
    n_ticks = int(os.environ.get("CHIP_STRESS_TICKS", "200000"))
    # two prefix lengths: 10k rows (cache-sized: half of every CU's workgroup slots, 256 workgroups) and 20k rows (512 workgroups)
    tick_sets = [[10_050 + 3 * j for j in range(3)], [20_150 + 3 * j for j in range(3)]]
    plants = [(tick_sets[0][0] - 1 - j, 5000 - j, 1) for j in range(3)] + [(tick_sets[1][1] - 1 - j, 15_000 - j, 1) for j in range(3)]
    p = capi.default_dot_params()
    p.min_new = -(1 << 30)                                   # every tick runs, whatever the previous l was
    monkeypatch.setenv("CHIP_TICK_FUSED", "0")
    with capi.Chip(D, capacity_hint=n_rows) as ref:
        ref.append_synthetic(n_rows, seed, plants)
        want = {l: bytes(ref.loop_tick(l, p)) for ls in tick_sets for l in ls}
    assert len(set(want.values())) == 6
    found = [capi.TickResult.from_buffer_copy(w).found for w in want.values()]
    assert sum(found) == 2
    monkeypatch.delenv("CHIP_TICK_FUSED")
    W = 16
    with capi.Chip(D, capacity_hint=n_rows) as chip:
        chip.append_synthetic(n_rows, seed, plants)
        for ls, share in zip(tick_sets, (0.6, 0.4)):
            n = int(n_ticks * share)
            pending = []
            bad = 0
            for i in range(n):
                if len(pending) == W:
                    s, l = pending.pop(0)
                    bad += bytes(chip.loop_tick_collect(s)) != want[l]
                l = ls[i % 3]
                chip.loop_tick_enqueue(l, i % W, p)
                pending.append((i % W, l))
            for s, l in pending:
                bad += bytes(chip.loop_tick_collect(s)) != want[l]
            assert bad == 0, (ls, bad, n)
        # ... and one tick at a time (the live system's mode): the host collects by POLLING the slot's completion word, which the
        # last workgroup stores with a system-scope release AFTER the record (round 5) -- a record read too early would differ
        bad = 0
        n_sync = max(3000, n_ticks // 8)
        for i in range(n_sync):
            l = tick_sets[i % 2][i % 3]
            bad += bytes(chip.loop_tick(l, p)) != want[l]
        assert bad == 0, (bad, n_sync)
