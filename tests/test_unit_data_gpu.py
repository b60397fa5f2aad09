"""GPU tests of the unit-L2 form of the synthetic generator (chip_db_append_synthetic_unit, ABI 7) -- SURVEY.md 8d's data: "rows =
unit-L2-norm", what NetVLAD's last layer emits (scripts/predict_utils.py:59-61 of the reference) -- and of the scan path on it.
bench.py's headline database is made by this call; these tests pin (a) the device rows against the oracle's definition bit for bit,
(b) ticks / top-k lists on such rows against the CPU oracle at 100k rows and at the headline size."""
import os

import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi

pytestmark = pytest.mark.gpu
SEED = 20190412


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("D,storage", [(4096, None), (8192, None), (256, None), (4096, "f64")])
def test_device_unit_generator_bit_identical_to_spec(D, storage):
    N = 3000
    plants = [(100, 5, 1), (101, 6, 2), (2999, 2000, 1)]
    with capi.Chip(D, storage=storage) as chip:
        chip.append_synthetic(1000, SEED, [p for p in plants if p[0] < 1000], unit=True)
        chip.append_synthetic(N - 1000, SEED, [p for p in plants if p[0] >= 1000], unit=True)   # appended in two calls
        rows = [0, 1, 5, 6, 100, 101, 999, 1000, 2000, 2999]
        want = oracle_lib.synth_rows(SEED, rows, D, plants, unit=True)
        if storage == "f64":
            assert chip.info()["storage_bytes"] == 8
            assert chip.read_rows_f64(rows).tobytes() == want.astype(np.float64).tobytes()   # double rows hold the float values
        else:
            assert chip.read_rows(rows).tobytes() == want.tobytes()
        norms = np.linalg.norm(want.astype(np.float64), axis=1)
        assert np.all(np.abs(norms - 1.0) < 2e-7)


def test_unit_and_plain_generators_interleave_in_one_db():
    """the two calls append to the same DB: every row is the form its call asked for"""
    D = 1024
    with capi.Chip(D) as chip:
        chip.append_synthetic(300, 7, [(10, 2, 1)])
        chip.append_synthetic(300, 7, [(310, 2, 2)], unit=True)
        rows = [2, 10, 299, 300, 310, 599]
        got = chip.read_rows(rows)
        assert got[:3].tobytes() == oracle_lib.synth_rows(7, rows[:3], D, [(10, 2, 1)]).tobytes()
        assert got[3:].tobytes() == oracle_lib.synth_rows(7, rows[3:], D, [(310, 2, 2)], unit=True).tobytes()


@pytest.mark.parametrize("G", [2, 8])
def test_unit_generator_in_a_group(G):
    """row % G shards on G sub-contexts of one device: the group's rows and its ticks are the single context's"""
    D, N = 1024, 2100
    plants, loops, _ = scenarios.loop_plants(N, 4, seed=31)
    with capi.Chip(D, devices=[0] * G) as grp, capi.Chip(D) as one:
        for c in (grp, one):
            c.append_synthetic(N, 5, plants, unit=True)
        rows = [0, 1, N // 2, N - 1] + [d for d, _, _ in plants[:4]]
        assert grp.read_rows(rows).tobytes() == oracle_lib.synth_rows(5, rows, D, plants, unit=True).tobytes()
        for l in scenarios.default_schedule(N):
            a, b = grp.loop_tick(l).as_dict(), one.loop_tick(l).as_dict()
            assert a == b and [float(x).hex() for x in a["maxv"]] == [float(x).hex() for x in b["maxv"]]


def test_unit_100k_full_oracle_parity():
    """4096-D x 100k unit rows: full CPU-oracle scan (threads) vs one GPU tick and the top-8 lists, bit-exact; the later exact
    duplicate of the revisited row wins the tie (Cerebro.cpp:1039-1043)."""
    D, N = 4096, 100_053
    l = N
    q, p = l - 1, 41234
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 5, p, 2)]
    with capi.Chip(D, capacity_hint=N) as chip:
        chip.append_synthetic(N, SEED, plants, unit=True)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(SEED, [l - 1, l - 2, l - 3], D, plants, unit=True)
        assert chip.read_rows([l - 1, l - 2, l - 3]).tobytes() == qrows.tobytes()
        wsc, wix = oracle_lib.scan_topk_synth(SEED, l - 50, D, qrows, 8, plants, nthreads=os.cpu_count() or 1, unit=True)
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1
        assert list(r.argmax) == list(wix[:, 0]) and r.idx_prev == p + 5 and r.idx_curr == q
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        assert 0.97 < r.maxv[0] < 0.99                    # unit rows: the score IS the cosine, 5 / sqrt(26)
        gs, gi = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(gi, wix) and np.array_equal(bits(gs), bits(wsc))
        # query 0 meets row p and its duplicate p + 5, queries 1 / 2 rows p - 1 / p - 2; nothing else comes near the threshold (N(0, 1/64))
        assert np.all(gs[0, :2] > 0.97) and np.all(gs[1:, 0] > 0.97)
        assert np.all(np.abs(gs[0, 2:]) < 0.2) and np.all(np.abs(gs[1:, 1:]) < 0.2)


def test_unit_1M_headline_size_parity():
    """BASELINE's headline size on 8d's data: 4096-D x 1M unit rows, one tick + the top-8 lists against the full threaded CPU-oracle
    scan, bit-exact; tie rule with an earlier and a later duplicate."""
    D, N = 4096, 1_000_053
    l = N
    q, p = l - 1, 777_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2), (123_456, p - 1, 2)]
    ncpu = os.cpu_count() or 1
    with capi.Chip(D, capacity_hint=N) as chip:
        chip.append_synthetic(N, SEED, plants, unit=True)
        r = chip.loop_tick(l)
        qrows = oracle_lib.synth_rows(SEED, [l - 1, l - 2, l - 3], D, plants, unit=True)
        assert chip.read_rows([l - 1, l - 2, l - 3]).tobytes() == qrows.tobytes()
        wsc, wix = oracle_lib.scan_topk_synth(SEED, l - 50, D, qrows, 8, plants, nthreads=min(ncpu, 128), unit=True)
        assert list(r.argmax) == list(wix[:, 0]) == [p + 4, p - 1, p - 2] and r.found == 1 and r.idx_prev == p + 4
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        gs, gi = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(gi, wix) and np.array_equal(bits(gs), bits(wsc))
        # a sample of rows across the DB (segment boundaries included) against the definition
        rows = [0, 16383, 16384, 500_000, 777_777, 777_781, 999_999, N - 1]
        assert chip.read_rows(rows).tobytes() == oracle_lib.synth_rows(SEED, rows, D, plants, unit=True).tobytes()
