"""Row N4: many-query batched mode (fp32 MFMA GEMM + fused exact top-k) vs the fp32 fmaf-chain oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi

pytestmark = pytest.mark.gpu


def check(chip, db, k, q, K):
    want_s, want_i = oracle_lib.scan_topk_fmaf(db, k, q, K)
    got_s, got_i = chip.query_batch(k, q, K)
    assert np.array_equal(got_i, want_i), (got_i[:2], want_i[:2])
    assert np.array_equal(got_s.view(np.uint32), want_s.astype(np.float32).view(np.uint32))     # bit-exact fp32 scores


# Q -> padded to 128s; a multiple of 256 takes the 256 x 256 / 8-wave tile, anything else the 128 x 128 / 4-wave one
@pytest.mark.parametrize("D,N,Q", [(32, 300, 5), (512, 1500, 64), (1024, 3000, 200), (4096, 2000, 130), (256, 2600, 300), (64, 1500, 512)])
def test_batch_parity(D, N, Q):
    plants, loops, ties = scenarios.loop_plants(N, 4, seed=D + Q)
    db = scenarios.build_db(7 * D, N, D, plants)
    rng = np.random.default_rng(Q)
    q = np.concatenate([db[rng.choice(N, Q - 2, replace=False)], oracle_lib.synth_rows(99, [1, 2], D)])
    q[0] = db[loops[0][1]]
    with capi.Chip(D) as chip:
        chip.append_f32(db)
        for K in (1, 8, 16):
            for k in (0, 1, 127, 128, 129, 255, 256, 257, N - 50, N):
                check(chip, db, k, q, K)
        if ties:
            s, t1, t2 = ties[0]
            sc, ix = chip.query_batch(N, db[[s]], 3)
            assert list(ix[0]) == [t2, t1, s] and sc[0][0] == sc[0][1] == sc[0][2]           # index-descending tie rule
        # fp32 MFMA scores agree with the fp64 scan of the tick path to fp32 round-off, and select the same best match
        s64, i64 = chip.query_rows(N - 50, [loops[0][1]], 1)
        s32, i32 = chip.query_batch(N - 50, db[[loops[0][1]]], 1)
        assert i32[0, 0] == i64[0, 0] and abs(float(s32[0, 0]) - s64[0, 0]) < 1e-5


def test_batch_errors():
    with capi.Chip(36) as chip:                      # D % 32 != 0
        chip.append_f32(np.zeros((4, 36), dtype=np.float32))
        with pytest.raises(capi.ChipError) as e:
            chip.query_batch(4, np.zeros((2, 36), dtype=np.float32), 4)
        assert e.value.status == capi.CHIP_ERR_UNSUPPORTED
    with capi.Chip(64) as chip:
        chip.append_f32(np.zeros((4, 64), dtype=np.float32))
        with pytest.raises(capi.ChipError) as e:
            chip.query_batch(5, np.zeros((2, 64), dtype=np.float32), 4)
        assert e.value.status == capi.CHIP_ERR_RANGE


def test_batch_sharded_lists_are_consistent():
    D, N, Q, G = 256, 1100, 70, 3
    db = scenarios.build_db(5, N, D, [])
    q = db[:Q]
    want_s, want_i = oracle_lib.scan_topk_fmaf(db, N, q, 8)
    parts = []
    for r in range(G):
        with capi.Chip(D, shard_rank=r, shard_count=G) as chip:
            chip.append_f32(db)
            parts.append(chip.query_batch(N, q, 8))
    for qi in range(Q):
        cand = sorted(((float(s), int(i)) for ps, pi in parts for s, i in zip(ps[qi], pi[qi]) if i >= 0), key=lambda t: (-t[0], -t[1]))[:8]
        assert [c[1] for c in cand] == list(want_i[qi])


# 5000 rows = 20 tiles of 256 / 40 tiles of 128.  The last, partial round of tiles is cut into query halves when that fills
# the grid better (2 * (tiles % wgs) <= wgs): (256, 8): 16 whole tiles + 4 x 2 halves; (100, 16): 32 + 8 x 2; (512, 6): 18 + 2 x 2
@pytest.mark.parametrize("Q,wgs", [(256, 1), (256, 3), (100, 2), (600, 2), (256, 8), (100, 16), (512, 6), (256, 19)])
def test_batch_claimed_tiles(Q, wgs, monkeypatch):
    """DB tiles are claimed from a counter by the workgroups of a query tile; with the normal grid a small DB gives every
    workgroup one tile, so cap the grid (CHIP_BATCH_WGS) to make each workgroup walk many claimed tiles, for one and for several
    query tiles (one counter each) and both tile shapes."""
    D, N = 128, 5000
    db = scenarios.build_db(11, N, D, [])
    rng = np.random.default_rng(Q + wgs)
    q = db[rng.choice(N, Q, replace=False)]
    monkeypatch.setenv("CHIP_BATCH_WGS", str(wgs))
    with capi.Chip(D) as chip:
        chip.append_f32(db)
        for k in (N, N - 257, 4097, 1023):
            check(chip, db, k, q, 8)
        check(chip, db, N, q[:40], 16)


@pytest.mark.parametrize("G,D,N,Q", [(8, 512, 9000, 256), (3, 128, 4100, 70), (2, 4096, 1500, 130)])
def test_batch_on_a_group_ctx(G, D, N, Q):
    """The many-query mode where the DB is row-sharded over the devices of a chip_create_multi ctx (round 3: CHIP_ERR_UNSUPPORTED):
    one db_gemm_topk pass per device over the rows it owns, the per-device lists merged on devices[0] -- bit for bit the result of one
    device holding the whole DB (indices and fp32 scores vs orc_scan_topk_fmaf_f32), at every prefix that cuts through the shards."""
    plants, loops, ties = scenarios.loop_plants(N, 4, seed=G + Q)
    db = scenarios.build_db(3 * D + G, N, D, plants)
    rng = np.random.default_rng(G * Q)
    q = db[rng.choice(N, Q, replace=False)]
    q[0] = db[loops[0][1]]
    with capi.Chip(D, devices=[0] * G) as chip:
        chip.append_f32(db[:N // 2])
        chip.append_f64(db[N // 2:].astype(np.float64))
        for K in (1, 8, 16):
            for k in (0, 1, G - 1, G, G + 1, 257, N - 50, N):
                check(chip, db, k, q, K)
        if ties:
            s, t1, t2 = ties[0]                         # the duplicates live on different devices: index-descending tie rule across shards
            sc, ix = chip.query_batch(N, db[[s]], 3)
            assert list(ix[0]) == [t2, t1, s] and sc[0][0] == sc[0][1] == sc[0][2]
        with pytest.raises(capi.ChipError) as e:
            chip.query_batch(N + 1, q, 4)
        assert e.value.status == capi.CHIP_ERR_RANGE
        # the tick path of the same group still works between batch calls (shared scan streams / query buffers)
        orc = oracle_lib.LoopOracle(db)
        for l in scenarios.default_schedule(N)[-20:]:
            chip.loop_reset()
            orc.state.last_l = 0
            g_, o_ = chip.loop_tick(l).as_dict(), orc.tick(l)
            assert (g_["found"], g_["argmax"]) == (o_["found"], o_["argmax"])
        check(chip, db, N, q, 8)


def test_batch_on_a_sharded_ctx_with_the_in_library_exchange():
    """One process per GPU layout at world size 1 (RCCL refuses two ranks on one device): local pass -> ncclAllGather of the
    [Q][topk] lists -> merge, through chip_query_batch_f32 on the sharded ctx."""
    D, N, Q = 256, 2100, 140
    db = scenarios.build_db(17, N, D, [])
    q = db[np.random.default_rng(3).choice(N, Q, replace=False)]
    with capi.Chip(D) as chip:
        chip.comm_init_rank(capi.comm_unique_id(), 1, 0)
        assert chip.info()["exchange"] == capi.CHIP_EXCHANGE_RCCL
        chip.append_f32(db)
        for K in (1, 8):
            for k in (0, 129, N):
                check(chip, db, k, q, K)
        with pytest.raises(capi.ChipError) as e:
            chip.query_batch(N + 5, q, 8)               # beyond what this rank has published: the failure mark, not a hang
        assert e.value.status == capi.CHIP_ERR_SHARD_FAILED
        check(chip, db, N, q, 8)                         # and the communicator is still in step
