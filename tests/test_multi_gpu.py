"""Multi-GPU INSIDE the C ABI (include/cerebro_hip.h "multi-GPU inside the library").  On a 1-GPU box:
  * chip_create_multi with the device list [0]*G: the full G-way code path (G sub-contexts with their own streams, row shards,
    replicated query ring, one host worker thread each, list exchange, merge + decision on the root) -- the exchange is the
    device-copy transport because RCCL refuses two ranks on one device;
  * chip_create_multi([0]) and chip_comm_init_rank(world 1): the same paths with the RCCL communicator (ncclCommInitAll /
    ncclCommInitRank, ncclAllGather enqueued in-stream between local and global merge) at world size 1.
  * BASELINE config 4 at FULL size (4096-D x 1M rows, 8 shards) through the same one-device group, bit-exact vs the CPU oracle.
On a box with >= 2 GPUs (skipped otherwise -- they light up the moment the suite runs on a multi-GPU node):
  * chip_create_multi(devices = 0..n-1): ncclCommInitAll over DISTINCT devices, ncclAllGather between the merges;
  * one process per GPU: chip_comm_unique_id / chip_comm_init_rank under 2 real processes with the RCCL exchange;
  both assert chip_get_info().comm_ranks == number of GPUs, i.e. that RCCL itself carried the exchange.
The caller-visible surface is the single-GPU one: chip_db_append_*, chip_loop_tick[_enqueue/_collect], chip_query_*, PnP."""
import os
import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi

pytestmark = pytest.mark.gpu


def same_tick(g, o):
    g = g.as_dict() if hasattr(g, "as_dict") else g
    for key in ("status", "found", "idx_curr", "idx_prev", "argmax"):
        assert g[key] == o[key], (key, g, o)
    assert float(g["score"]).hex() == float(o["score"]).hex()
    assert [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_group_ticks_bit_exact_vs_oracle(G):
    D, N = 1024, 1600
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=40 + G)
    db = scenarios.build_db(4000 + G, N, D, plants)
    orc = oracle_lib.LoopOracle(db)
    with capi.Chip(D, devices=[0] * G, copy_exchange=True) as chip:
        info = chip.info()
        assert info["n_devices"] == G and info["exchange"] == capi.CHIP_EXCHANGE_COPY and info["shard_count"] == G
        assert chip.append_f64(db[:700].astype(np.float64)) == 0
        assert chip.append_f32(db[700:]) == 700
        assert chip.size() == N and chip.info()["rows_local"] == len(range(0, N, G))
        # irregular schedule through the synchronous entry point
        sched = [1, 3, 5, 30, 55, 57, 58, 61] + list(range(64, N + 1, 3))
        ls = set(sched)
        for l, _, _ in loops:
            ls -= {l - 1, l - 2}
            ls.add(l)
        n_found = 0
        for l in sorted(ls):
            o = orc.tick(l)
            same_tick(chip.loop_tick(l), o)
            assert chip.last_l() == orc.state.last_l
            n_found += o["found"]
        assert n_found >= len(loops)
        # pipelined form: 24 ticks in flight, collected in order
        chip.loop_reset()
        orc2 = oracle_lib.LoopOracle(db)
        sched = scenarios.default_schedule(N)
        for base in range(0, len(sched), 24):
            chunk = sched[base:base + 24]
            for s, l in enumerate(chunk):
                chip.loop_tick_enqueue(l, s)
            for s, l in enumerate(chunk):
                same_tick(chip.loop_tick_collect(s), orc2.tick(l))
        with pytest.raises(capi.ChipError):
            chip.loop_tick_collect(0)
        # top-k queries by row and by external vector, every (nq, K)
        rows = [N - 1, N - 2, N - 3, loops[0][1]]
        for nq in (1, 3, 4):
            for K in (1, 5, 16):
                for k in (0, 1, 7, N - 50, N):
                    want = oracle_lib.scan_topk(db, k, db[rows[:nq]], K)
                    got = chip.query_rows(k, rows[:nq], K)
                    assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        q = oracle_lib.synth_rows(5, [10, 11, 12], D)
        got, want = chip.query_vectors(N, q, 8), oracle_lib.scan_topk(db, N, q, 8)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        # tie rule across shards: the duplicates of one row live on different devices
        s, t1, t2 = ties[0]
        l, qrow, p = loops[0]
        sc, ix = chip.query_rows(l - 50, [qrow], 3)
        assert list(ix[0]) == [t2, t1, s] and sc[0][0] == sc[0][1] == sc[0][2]
        # full score vector and row read-back are assembled from all shards
        u = chip.query_scores(N - 50, N - 1)
        assert np.array_equal(bits(u), bits(oracle_lib.scores(db, N - 50, db[N - 1])))
        back = chip.read_rows(list(range(0, N, 97)) + [N - 1])
        assert back.tobytes() == db[list(range(0, N, 97)) + [N - 1]].tobytes()
        # error paths keep their status codes
        with pytest.raises(capi.ChipError) as e:
            chip.loop_tick(N + 1)
        assert e.value.status == capi.CHIP_ERR_RANGE
        with pytest.raises(capi.ChipError) as e:
            chip.set_stream(0)
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED


def test_group_8way_4096d_growth_and_pose():
    """BASELINE config 4's layout (8 shards, 4096-D) on one device: on-device generator, appends that outrun the ring, and
    the pose verifier through the same ctx (it runs on devices[0])."""
    import np_mirror_pnp as M
    D, N, seed = 4096, 20_053, 20190412
    l = N
    q, p = l - 1, 7_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 5, p, 2)]
    with capi.Chip(D, capacity_hint=N, devices=[0] * 8) as chip:      # a repeated device implies the copy exchange
        assert chip.info()["exchange"] == capi.CHIP_EXCHANGE_COPY
        chip.append_synthetic(N, seed, plants)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, plants)
        wsc, wix = oracle_lib.scan_topk_synth(seed, l - 50, D, qrows, 8, plants, nthreads=8)
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1 and r.idx_prev == p + 5 and r.idx_curr == q
        assert list(r.argmax) == list(wix[:, 0]) and [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        got = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(got[1], wix) and np.array_equal(bits(got[0]), bits(wsc))
        # ANY appended row can be a query row of chip_query_rows / chip_query_scores: rows that have left the replicated ring are
        # fetched from the sub-context that owns them (round 2 returned CHIP_ERR_RANGE here)
        old_rows = [5, 4098, p]
        oq = oracle_lib.synth_rows(seed, old_rows, D, plants)
        for k1 in (100, 9_000, l - 50):
            wsc2, wix2 = oracle_lib.scan_topk_synth(seed, k1, D, oq, 5, plants, nthreads=8)
            got2 = chip.query_rows(k1, old_rows, 5)
            assert np.array_equal(got2[1], wix2) and np.array_equal(bits(got2[0]), bits(wsc2))
        u = chip.query_scores(3000, 5)
        db3k = oracle_lib.synth_rows(seed, range(3000), D, plants)
        assert np.array_equal(bits(u), bits(oracle_lib.scores(db3k, 3000, oq[0])))
        with pytest.raises(capi.ChipError) as e:
            chip.query_rows(100, [N], 1)           # not an appended row
        assert e.value.status == capi.CHIP_ERR_RANGE
        X, uv, T, inl = M.make_scene(N=200, outlier_frac=0.2, noise_px=0.5, seed=7)
        prm = capi.default_ransac_params(); prm.n_hypotheses = 64; prm.seed = 7
        g = chip.pnp_ransac(X, uv, prm)
        o = oracle_lib.pnp_ransac(X, uv, oracle_lib.ransac_params(n_hypotheses=64, seed=7))
        assert g["summary"]["best_hypothesis"] == o["summary"]["best_hypothesis"] >= 0 and np.array_equal(g["mask"], o["mask"])


def _tick_parity(chip, db, N):
    orc = oracle_lib.LoopOracle(db)
    for l in [2, 40] + scenarios.default_schedule(N):
        same_tick(chip.loop_tick(l), orc.tick(l))
    chip.loop_reset()
    orc2 = oracle_lib.LoopOracle(db)
    sched = scenarios.default_schedule(N)
    for base in range(0, len(sched), 20):
        chunk = sched[base:base + 20]
        for s, l in enumerate(chunk):
            chip.loop_tick_enqueue(l, s)
        for s, l in enumerate(chunk):
            same_tick(chip.loop_tick_collect(s), orc2.tick(l))


def test_group_over_rccl_world1():
    """chip_create_multi([0]) without the copy flag: ncclCommInitAll over one device, ncclAllGather between the merges."""
    D, N = 1024, 900
    plants, loops, _ = scenarios.loop_plants(N, 4, seed=3)
    db = scenarios.build_db(5, N, D, plants)
    with capi.Chip(D, devices=[0]) as chip:
        assert chip.info()["exchange"] == capi.CHIP_EXCHANGE_RCCL
        chip.append_f32(db)
        _tick_parity(chip, db, N)
        got, want = chip.query_rows(N - 50, [N - 1, N - 2], 5), oracle_lib.scan_topk(db, N - 50, db[[N - 1, N - 2]], 5)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))


def test_comm_init_rank_world1():
    """One process per GPU layout at world size 1: chip_comm_unique_id -> chip_comm_init_rank, after which chip_loop_tick* on
    the sharded ctx run scan -> local merge -> ncclAllGather -> merge + decision inside the library."""
    D, N = 512, 1000
    plants, loops, _ = scenarios.loop_plants(N, 4, seed=13)
    db = scenarios.build_db(15, N, D, plants)
    with capi.Chip(D) as chip:
        uid = capi.comm_unique_id()
        assert len(uid) == capi.CHIP_COMM_ID_BYTES
        with pytest.raises(capi.ChipError) as e:
            chip.comm_init_rank(uid, 2, 0)          # does not match the shard layout of chip_create
        assert e.value.status == capi.CHIP_ERR_INVALID_ARG
        chip.comm_init_rank(uid, 1, 0)
        assert chip.info()["exchange"] == capi.CHIP_EXCHANGE_RCCL
        chip.append_f32(db)
        _tick_parity(chip, db, N)
        got, want = chip.query_rows(N - 50, [N - 1], 8), oracle_lib.scan_topk(db, N - 50, db[[N - 1]], 8)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        with pytest.raises(capi.ChipError) as e:
            chip.comm_init_rank(uid, 1, 0)          # already attached
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED


def test_sharded_ctx_without_exchange_still_refuses_ticks():
    with capi.Chip(256, shard_rank=1, shard_count=2) as chip:
        chip.append_f32(oracle_lib.synth_rows(1, range(100), 256))
        with pytest.raises(capi.ChipError) as e:
            chip.loop_tick(100)
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED      # host-driven exchange: chip_scan_local + chip_merge_decide


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def test_group_8way_1M_full_size_parity():
    """BASELINE config 4's workload at its own size -- 4096-D x 1 000 053 rows sharded 8-way -- through chip_create_multi on ONE
    device (8 sub-contexts, row i on sub-context i % 8, replicated query ring, device-copy exchange): the tick, the top-8 lists
    and prefix queries are bit-exact vs the full CPU-oracle scan, exactly as test_1M_full_size_parity_and_properties checks the
    unsharded ctx."""
    D, N, seed = 4096, 1_000_053, 20190412
    l = N
    q, p = l - 1, 777_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2), (123_456, p - 1, 2)]
    ncpu = os.cpu_count() or 1
    with capi.Chip(D, capacity_hint=N, devices=[0] * 8) as chip:
        info = chip.info()
        assert info["n_devices"] == 8 and info["exchange"] == capi.CHIP_EXCHANGE_COPY and info["comm_ranks"] == 0
        chip.append_synthetic(N, seed, plants)
        assert chip.size() == N and chip.info()["rows_local"] == len(range(0, N, 8))
        r = chip.loop_tick(l)
        wsc, wix = scenarios.cached_scan_topk_synth(seed, l - 50, D, [l - 1, l - 2, l - 3], 8, plants, nthreads=min(ncpu, 128))
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1 and r.idx_curr == q and r.idx_prev == p + 4
        assert list(r.argmax) == list(wix[:, 0]) == [p + 4, p - 1, p - 2]
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        got = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(got[1], wix) and np.array_equal(bits(got[0]), bits(wsc))
        # prefixes that cut through the shards at every residue
        for k1 in (8, 1001, 123_457, 500_003, p + 5):
            sc, ix = chip.query_rows(k1, [l - 1, l - 2, l - 3], 8)
            assert np.all(ix < k1) and np.all(np.diff(sc, axis=1) <= 0)
            for qi in range(3):
                keep = [(s, i) for s, i in zip(wsc[qi], wix[qi]) if i < k1]
                for (s, i), gs, gi in zip(keep, sc[qi], ix[qi]):
                    assert (s, i) == (gs, gi)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: ncclCommInitAll over distinct devices")
def test_group_over_rccl_distinct_devices():
    """chip_create_multi over the GPUs of this node (up to 8): the RCCL exchange with one rank per device."""
    n = min(_n_gpus(), 8)
    D, N = 1024, 2400
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=77)
    db = scenarios.build_db(4100, N, D, plants)
    with capi.Chip(D, devices=list(range(n))) as chip:
        info = chip.info()
        assert info["n_devices"] == n and info["exchange"] == capi.CHIP_EXCHANGE_RCCL and info["comm_ranks"] == n
        chip.append_f32(db[:1000])
        chip.append_f64(db[1000:].astype(np.float64))
        _tick_parity(chip, db, N)
        for nq, K in ((1, 1), (3, 8), (4, 16)):
            rows = [N - 1, N - 2, N - 3, loops[0][1]][:nq]
            got, want = chip.query_rows(N - 50, rows, K), oracle_lib.scan_topk(db, N - 50, db[rows], K)
            assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        u = chip.query_scores(N - 50, N - 1)
        assert np.array_equal(bits(u), bits(oracle_lib.scores(db, N - 50, db[N - 1])))
    # BASELINE config 4's shard size on real devices: 125k rows per GPU when n == 8
    D, N, seed = 4096, 125_000 * n + 53, 20190412
    l = N
    q, p = l - 1, N // 2
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2)]
    with capi.Chip(D, capacity_hint=N, devices=list(range(n))) as chip:
        assert chip.info()["comm_ranks"] == n
        chip.append_synthetic(N, seed, plants)
        r = chip.loop_tick(l)
        wsc, wix = scenarios.cached_scan_topk_synth(seed, l - 50, D, [l - 1, l - 2, l - 3], 8, plants, nthreads=min(os.cpu_count() or 1, 128))
        assert r.found == 1 and r.idx_prev == p + 4 and list(r.argmax) == list(wix[:, 0])
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]


def _rccl_rank_worker(rank, world, uid_path, ret):
    import time
    import torch
    from cerebro_amd import capi as capi_
    torch.cuda.set_device(rank)
    D, N = 1024, 1800
    plants, loops, ties = scenarios.loop_plants(N, 5, seed=21)
    db = scenarios.build_db(733, N, D, plants)
    with capi_.Chip(D, device=rank, shard_rank=rank, shard_count=world) as chip:
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(capi_.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        chip.comm_init_rank(open(uid_path, "rb").read(), world, rank)
        info = chip.info()
        assert info["exchange"] == capi_.CHIP_EXCHANGE_RCCL and info["comm_ranks"] == world
        chip.append_f32(db)
        orc = oracle_lib.LoopOracle(db)
        n_found = 0
        for l in scenarios.default_schedule(N):          # collective calls: every rank makes them in the same order
            g, o = chip.loop_tick(l).as_dict(), orc.tick(l)
            for key in ("status", "found", "idx_curr", "idx_prev", "argmax"):
                assert g[key] == o[key], (rank, l, key, g, o)
            assert [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]
            n_found += g["found"]
        got, want = chip.query_rows(N - 50, [N - 1, N - 2], 8), oracle_lib.scan_topk(db, N - 50, db[[N - 1, N - 2]], 8)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        ret[rank] = n_found


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
def test_comm_init_rank_real_processes_over_rccl(tmp_path):
    """One process per GPU (BASELINE config 4's launch shape): ncclCommInitRank inside the library, in-stream ncclAllGather."""
    import torch.multiprocessing as mp
    world = min(_n_gpus(), 8)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_rank_worker, args=(world, str(tmp_path / "uid.bin"), ret), nprocs=world, join=True)
    assert len(ret) == world and len(set(ret.values())) == 1 and next(iter(ret.values())) > 0


@pytest.mark.parametrize("G", [3, 8])
def test_group_bulk_append_owner_only(G):
    """Bulk appends that outrun the ring (the cold start after loadStateFromDisk, Cerebro.cpp:133-161,1005): every device is sent
    only the rows it owns (+ the newest CHIP_RING_ROWS for its ring); rows read back from their owners and the ticks over them are
    those of the oracle, for float64 and float32 wire types and batch starts at every residue."""
    D = 256
    sizes = [5000, 1, 4097, 7, 9001, 2]                 # > ring, tiny, ring + 1, ...
    N = sum(sizes)
    plants, loops, ties = scenarios.loop_plants(N, 8, seed=11 + G)
    db = scenarios.build_db(900 + G, N, D, plants)
    with capi.Chip(D, devices=[0] * G) as chip:
        at = 0
        for i, m in enumerate(sizes):
            blk = db[at:at + m]
            assert (chip.append_f64(blk.astype(np.float64)) if i % 2 == 0 else chip.append_f32(blk)) == at
            at += m
            assert chip.size() == at and chip.info()["rows_local"] == len(range(0, at, G))
        rows = list(range(0, N, 613)) + [4095, 4096, 4097, 5000, 5001, N - 1]
        assert chip.read_rows(rows).tobytes() == db[rows].tobytes()
        # live ticks read their query rows from the replicated ring (they trail the append head by < CHIP_RING_ROWS - 3 rows) ...
        sched = [l for l in scenarios.default_schedule(N) if l >= N - 4000]
        sched = sorted(set(sched[::5]) | {lp[0] for lp in loops if lp[0] >= N - 4000})
        orc = oracle_lib.LoopOracle(db)
        for l in sched:
            same_tick(chip.loop_tick(l), orc.tick(l))
        # ... and a tick further back than the ring fetches them from the sub-contexts that own them: the WHOLE schedule of a run can
        # be replayed over the cold-started group (cerebro_replay --state --devices ...; round 3 returned CHIP_ERR_RANGE here),
        # synchronously and pipelined across the ring boundary
        chip.loop_reset()
        orc = oracle_lib.LoopOracle(db)
        old = sorted(set(scenarios.default_schedule(N)[::9]) | {lp[0] for lp in loops})
        assert old[0] < N - 15000 and sum(1 for l in old if N - l > 4093) > 100
        for l in old:
            same_tick(chip.loop_tick(l), orc.tick(l))
        chip.loop_reset()
        orc = oracle_lib.LoopOracle(db)
        around = [l for l in scenarios.default_schedule(N) if N - 4200 <= l <= N - 3990]
        for base in range(0, len(around), 16):
            chunk = around[base:base + 16]
            for s_, l in enumerate(chunk):
                chip.loop_tick_enqueue(l, s_)
            for s_, l in enumerate(chunk):
                same_tick(chip.loop_tick_collect(s_), orc.tick(l))
        # a batch with a non-finite value is rejected as a whole on every device, whichever device owns the bad row
        bad = db[:G + 2].astype(np.float64).copy()
        bad[G, 3] = np.nan
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(bad)
        assert e.value.status == capi.CHIP_ERR_NONFINITE and chip.size() == N
        nf32 = db[:G + 2].astype(np.float64).copy()
        nf32[1, 0] = 0.1                                  # not float32-representable: the float DB already holds rows
        with pytest.raises(capi.ChipError) as e:
            chip.append_f64(nf32)
        assert e.value.status == capi.CHIP_ERR_NOT_F32 and chip.size() == N
        assert chip.append_f32(db[:3]) == N               # and the group is still usable


def test_group_auto_switch_to_double_is_one_decision():
    """An undecided group DB whose first batch holds ONE value that is not float32-representable -- owned by one device only --
    becomes a double-row DB on every device (the decision is taken from the OR of the devices' validation bits)."""
    D, N, G = 128, 40, 4
    db = oracle_lib.synth_rows(3, range(N), D).astype(np.float64)
    db[6, 17] = 0.1                                       # row 6 lives on device 2
    with capi.Chip(D, devices=[0] * G) as chip:
        assert chip.append_f64(db) == 0
        assert chip.info()["storage_bytes"] == 8
        back = chip.read_rows_f64(list(range(N))) if hasattr(chip, "read_rows_f64") else None
        if back is not None:
            assert back.tobytes() == db.tobytes()
        sc, ix = chip.query_rows(N, [6], 1)
        assert ix[0][0] == 6


def test_failed_shard_marks_the_tick_for_everyone(monkeypatch, hooks_lib):
    """A shard whose own validation fails must not leave a collective tick (the others have enqueued their exchange): it takes part
    with the marked neutral list, the merge reports the mark, the call fails with CHIP_ERR_SHARD_FAILED, last_l is as before, and
    the NEXT tick is correct again -- the exchange never goes out of step.  CHIP_TEST_FAIL_SHARD makes shard 2 fail every 5th call."""
    monkeypatch.setenv("CHIP_TEST_FAIL_SHARD", "2:5")
    D, N, G = 512, 1300, 4
    plants, loops, ties = scenarios.loop_plants(N, 5, seed=91)
    db = scenarios.build_db(77, N, D, plants)
    with capi.Chip(D, devices=[0] * G) as chip:
        chip.append_f32(db)
        orc = oracle_lib.LoopOracle(db)
        n_failed = n_ok = 0
        for l in scenarios.default_schedule(N):
            before = chip.last_l()
            try:
                g = chip.loop_tick(l)
            except capi.ChipError as e:
                assert e.status == capi.CHIP_ERR_SHARD_FAILED
                assert chip.last_l() == before             # the tick had no effect
                n_failed += 1
                g = chip.loop_tick(l)                       # retry: the shard takes part again
            same_tick(g, orc.tick(l))
            n_ok += 1
        assert n_failed >= 10 and n_ok == len(scenarios.default_schedule(N))
        # pipelined: the failed slot reports at collect time, the slots around it are untouched
        sched = scenarios.default_schedule(N)[:80]
        stateless = oracle_lib.LoopOracle(db)
        n_failed = 0
        for base in range(0, len(sched), 4):
            chunk = sched[base:base + 4]
            chip.loop_reset()
            for s_, l in enumerate(chunk):
                chip.loop_tick_enqueue(l, s_)
            last_ok = None
            for s_, l in enumerate(chunk):
                try:
                    g = chip.loop_tick_collect(s_)
                except capi.ChipError as e:
                    assert e.status == capi.CHIP_ERR_SHARD_FAILED
                    n_failed += 1
                    last_ok = False
                    continue
                last_ok = True
                stateless.state.last_l = 0
                same_tick(g, stateless.tick(l))
            # a failed tick rolls last_l back only while it is the newest one enqueued: the commit of a LATER tick that succeeded
            # stands (with ticks pipelined the old rollback erased it)
            if last_ok:
                assert chip.last_l() == chunk[-1]
        assert n_failed >= 10
        # queries carry the mark too
        seen = 0
        for _ in range(12):
            try:
                got = chip.query_rows(N - 50, [N - 1], 4)
                want = oracle_lib.scan_topk(db, N - 50, db[[N - 1]], 4)
                assert np.array_equal(got[1], want[1])
            except capi.ChipError as e:
                assert e.status == capi.CHIP_ERR_SHARD_FAILED
                seen += 1
        assert seen >= 1


@pytest.mark.parametrize("mode", ["hang", "fail"])
def test_rccl_bootstrap_under_a_deadline(monkeypatch, mode, hooks_lib):
    """ncclCommInitAll / ncclCommInitRank are blocking rendezvous; on a node where they cannot complete they hang.  The library runs
    them on a helper thread under CHIP_COMM_INIT_TIMEOUT_MS: a group falls back to the device-copy exchange (same answers), a
    sharded ctx gets CHIP_ERR_COMM, chip_get_info().comm_init_abandoned says a helper is still stuck.  CHIP_TEST_COMM_INIT makes the
    bootstrap hang / fail on a healthy box (and says so on stderr)."""
    import time
    monkeypatch.setenv("CHIP_TEST_COMM_INIT", mode)
    monkeypatch.setenv("CHIP_COMM_INIT_TIMEOUT_MS", "1500")
    D, N = 512, 900
    plants, loops, _ = scenarios.loop_plants(N, 3, seed=29)
    db = scenarios.build_db(31, N, D, plants)
    t0 = time.time()
    with capi.Chip(D, devices=[0]) as chip:                  # distinct devices -> RCCL transport is asked for
        assert time.time() - t0 < 60
        info = chip.info()
        assert info["exchange"] == capi.CHIP_EXCHANGE_COPY and info["comm_ranks"] == 0
        assert info["comm_init_abandoned"] == (1 if mode == "hang" else 0)
        assert ("did not return" in chip.last_comm_error()) == (mode == "hang")
        chip.append_f32(db)
        _tick_parity(chip, db, N)
    with capi.Chip(D) as chip:
        with pytest.raises(capi.ChipError) as e:
            chip.comm_init_rank(capi.comm_unique_id(), 1, 0)
        assert e.value.status == capi.CHIP_ERR_COMM
        assert chip.info()["comm_init_abandoned"] == (1 if mode == "hang" else 0) and chip.info()["exchange"] == capi.CHIP_EXCHANGE_NONE
        chip.append_f32(db)                                   # the ctx is an ordinary single-GPU ctx still
        _tick_parity(chip, db, N)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs: hipMemcpyPeerAsync between distinct devices")
def test_group_copy_exchange_distinct_devices():
    """CHIP_MULTI_EXCHANGE_COPY over DISTINCT devices: the fallback transport of a group whose RCCL bootstrap hung or failed -- per-device
    lists pulled into the root's gather buffer with hipMemcpyPeerAsync behind events, query rows fetched from their owners with peer
    copies.  (On one device the same code path runs with plain device copies; this is the cross-device leg.)"""
    n = min(_n_gpus(), 8)
    D, N = 1024, 2400
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=78)
    db = scenarios.build_db(4200, N, D, plants)
    with capi.Chip(D, devices=list(range(n)), copy_exchange=True) as chip:
        info = chip.info()
        assert info["n_devices"] == n and info["exchange"] == capi.CHIP_EXCHANGE_COPY and info["comm_ranks"] == 0
        chip.append_f32(db[:1000])
        chip.append_f64(db[1000:].astype(np.float64))
        _tick_parity(chip, db, N)
        for nq, K in ((1, 1), (3, 8), (4, 16)):
            rows = [N - 1, N - 2, N - 3, loops[0][1]][:nq]      # loops[0][1]: an old row, fetched from its owner by a peer copy
            got, want = chip.query_rows(N - 50, rows, K), oracle_lib.scan_topk(db, N - 50, db[rows], K)
            assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        u = chip.query_scores(N - 50, 7)
        assert np.array_equal(bits(u), bits(oracle_lib.scores(db, N - 50, db[7])))
    # BASELINE config 4's shard size over the copy exchange
    D, N, seed = 4096, 125_000 * n + 53, 20190412
    l = N
    q, p = l - 1, N // 2
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2)]
    with capi.Chip(D, capacity_hint=N, devices=list(range(n)), copy_exchange=True) as chip:
        chip.append_synthetic(N, seed, plants)
        r = chip.loop_tick(l)
        wsc, wix = scenarios.cached_scan_topk_synth(seed, l - 50, D, [l - 1, l - 2, l - 3], 8, plants, nthreads=min(os.cpu_count() or 1, 128))
        assert r.found == 1 and r.idx_prev == p + 4 and list(r.argmax) == list(wix[:, 0])
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]


def test_two_consecutive_failed_pipelined_ticks_unwind_to_the_last_good_l(monkeypatch, hooks_lib):
    """ticks A and B are both enqueued and both come back CHIP_TICK_FAILED.  A's collect cannot roll back (B was enqueued on
    top of it); B's collect must not restore A's l (a pass that never reached Cerebro.cpp:1098) but the last_l from before A."""
    monkeypatch.setenv("CHIP_TEST_FAIL_SHARD", "1:1")           # shard 1 fails EVERY collective call
    D, N = 256, 900
    db = scenarios.build_db(5, N, D, [])
    with capi.Chip(D, devices=[0, 0]) as chip:
        chip.append_f32(db)
        assert chip.last_l() == 0
        for depth in (2, 3):
            ls = [400 + 10 * i for i in range(depth)]
            for s_, l in enumerate(ls):
                chip.loop_tick_enqueue(l, s_)
            assert chip.last_l() == ls[-1]                       # provisional commits
            for s_ in range(depth):
                with pytest.raises(capi.ChipError) as ei:
                    chip.loop_tick_collect(s_)
                assert ei.value.status == capi.CHIP_ERR_SHARD_FAILED
            assert chip.last_l() == 0, depth                     # unwound through every failed predecessor
