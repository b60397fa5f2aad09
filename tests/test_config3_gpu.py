"""BASELINE config 3 as one scenario: synthetic 4096-D x 100k keyframe DB + 512-correspondence DlsPnpWithRansac with 1 000
hypotheses, on ONE ctx, the tick stream and the pose verifier running concurrently from two threads (dot_product_th and
loopcandidate_consumer_th of the reference, cerebro_node.cpp:499,509).  Ticks are checked against the planted schedule and the
full-oracle scan, the RANSAC result against the FROZEN fixture (tests/golden/pnp_golden.json, case 1000 hypotheses / seed 4242)."""
import json
import os
import threading
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
from cerebro_amd import capi

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_config3_ticks_and_pnp_share_the_gpu():
    D, rows, seed = 4096, 100_000, 20190412
    n_ticks = 24
    ls = [rows + 50 + 3 * i for i in range(n_ticks)]
    rng = np.random.default_rng(3)
    plants, expect = [], []
    for i, l in enumerate(ls):
        if i % 3 == 0:
            p = int(rng.integers(1000, rows - 1000))
            plants += [(l - 1 - j, p - j, 1) for j in range(3)]
            expect.append((l - 1, p))
        else:
            expect.append(None)
    g = json.loads((GOLD / "pnp_golden.json").read_text())
    case = [c for c in g["cases"] if c["n_hypotheses"] == 1000 and c["seed"] == 4242][0]
    X, uv = np.array(g["X"]), np.array(g["uv"])
    T_want = np.array([float.fromhex(x) for x in case["T_colmajor_hex"]]).reshape(4, 4).T
    errors, pnp_runs = [], [0]
    stop = threading.Event()
    with capi.Chip(D, capacity_hint=ls[-1]) as chip:
        chip.append_synthetic(ls[-1], seed, sorted(plants))

        def pnp_caller():
            try:
                prm = capi.default_ransac_params(); prm.n_hypotheses = 1000; prm.seed = 4242
                while not stop.is_set():
                    r = chip.pnp_ransac(X, uv, prm)
                    s = r["summary"]
                    assert (s["best_hypothesis"], s["n_models"], s["n_inliers"]) == (case["best_hypothesis"], case["n_models"], case["n_inliers"])
                    assert np.packbits(r["mask"]).tobytes().hex() == case["mask_hex"]                  # inlier mask bit-exact
                    assert np.linalg.norm(r["T"] - T_want) <= 1e-4 * np.linalg.norm(T_want)           # north-star pose tolerance
                    pnp_runs[0] += 1
            except Exception as e:  # pragma: no cover
                errors.append(e)

        th = threading.Thread(target=pnp_caller)
        th.start()
        try:
            results = []
            for rep in range(3):                      # the tick stream: pipelined, 8 in flight
                chip.loop_reset()
                pending = []
                for i, l in enumerate(ls):
                    if len(pending) == 8:
                        results.append(chip.loop_tick_collect(pending.pop(0)))
                    chip.loop_tick_enqueue(l, i % 8)
                    pending.append(i % 8)
                while pending:
                    results.append(chip.loop_tick_collect(pending.pop(0)))
        finally:
            stop.set()
            th.join(timeout=120)
        assert not errors, errors
        assert pnp_runs[0] >= 3
        for r, e in zip(results, expect * 3):
            if e is None:
                assert r.found == 0
            else:
                assert r.found == 1 and (r.idx_curr, r.idx_prev) == e
        # one tick against the full CPU oracle scan of the 100k prefix
        l = ls[0]
        q = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, sorted(plants))
        wsc, wix = oracle_lib.scan_topk_synth(seed, l - 50, D, q, 8, sorted(plants), nthreads=os.cpu_count() or 1)
        got = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(got[1], wix) and got[0].tobytes() == wsc.tobytes()


def test_group_ctx_appender_and_querier_threads():
    """The same thread contract on a chip_create_multi ctx (4 sub-contexts on device 0): an appender thread feeds all shards while
    the querier ticks; every tick must equal the oracle's answer for the prefix it saw."""
    import scenarios
    D, N = 512, 1800
    plants = []
    for q, p in [(500, 100), (1100, 400), (1700, 900)]:
        plants += [(q - j, p - j, 1) for j in range(12)]
    db = scenarios.build_db(6, N, D, sorted(plants))
    errors, found, last_tick = [], [], [0]
    stop = threading.Event()
    with capi.Chip(D, capacity_hint=64, devices=[0, 0, 0, 0]) as chip:
        def appender():
            try:
                for i in range(0, N, 3):
                    while i - last_tick[0] > 6 and not errors:
                        pass
                    chip.append_f64(db[i:i + 3].astype(np.float64))
            except Exception as e:  # pragma: no cover
                errors.append(e)
            finally:
                stop.set()

        def querier():
            try:
                last = 0
                while True:
                    done = stop.is_set()
                    l = chip.size()
                    if l - last >= 3:
                        r = chip.loop_tick(l)
                        if r.status != capi.CHIP_TICK_SKIPPED:
                            last = l
                            last_tick[0] = l
                        if r.found:
                            found.append((r.idx_curr, r.idx_prev, r.score))
                        if r.status == capi.CHIP_TICK_SCANNED and l % 5 == 0:
                            sc, ix = oracle_lib.scan_topk(db, l - 50, db[[l - 1, l - 2, l - 3]], 1)
                            assert list(r.argmax) == list(ix[:, 0]) and list(r.maxv) == list(sc[:, 0])
                    if done and chip.size() - last < 3:
                        break
            except Exception as e:  # pragma: no cover
                errors.append(e)

        ts = [threading.Thread(target=f) for f in (appender, querier)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errors, errors
        assert chip.size() == N and len(found) > 0
        for cur, prev, score in found:
            assert score == oracle_lib.dot_tree(db[cur], db[prev]) and score > 0.85
