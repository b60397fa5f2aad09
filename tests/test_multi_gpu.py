"""Multi-GPU INSIDE the C ABI (include/cerebro_hip.h "multi-GPU inside the library"), as far as a 1-GPU box allows:
  * chip_create_multi with the device list [0]*G: the full G-way code path (G sub-contexts with their own streams, row shards,
    replicated query ring, one host worker thread each, list exchange, merge + decision on the root) -- the exchange is the
    device-copy transport because RCCL refuses two ranks on one device;
  * chip_create_multi([0]) and chip_comm_init_rank(world 1): the same paths with the RCCL communicator (ncclCommInitAll /
    ncclCommInitRank, ncclAllGather enqueued in-stream between local and global merge) at world size 1.
The caller-visible surface is the single-GPU one: chip_db_append_*, chip_loop_tick[_enqueue/_collect], chip_query_*, PnP."""
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
        # the newest rows are query-able on every shard (ring), old rows are not query rows any more (documented limit)
        with pytest.raises(capi.ChipError) as e:
            chip.query_rows(100, [5], 1)
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
