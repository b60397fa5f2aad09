"""Thread-safety contract of the ctx (include/cerebro_hip.h): one appender thread (desc_th), one querier thread
(dot_product_th) and one PnP caller (loopcandidate_consumer_th) share a chip_ctx, as in cerebro_node.cpp:487-509."""
import threading
import time

import numpy as np
import pytest

import oracle_lib
import scenarios
from cerebro_amd import capi
from cerebro_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def test_appender_querier_pnp_threads_share_a_ctx():
    D, N = 512, 2400
    # revisits of 12 consecutive keyframes: tick positions depend on thread timing (as in the live system), so any l
    # whose three newest rows fall inside a revisit must fire
    plants = []
    for q, p in [(400, 120), (900, 300), (1500, 700), (2100, 1100)]:
        plants += [(q - j, p - j, 1) for j in range(12)]
    db = scenarios.build_db(4, N, D, sorted(plants))
    X, uv, T, inl = make_scene(N=300, outlier_frac=0.2, noise_px=0.5, seed=5)
    want_pnp = oracle_lib.pnp_ransac(X, uv, oracle_lib.ransac_params(seed=9))
    errors, found, pnp_runs, last_tick = [], [], [0], [0]
    stop = threading.Event()
    with capi.Chip(D, capacity_hint=64) as chip:          # tiny hint: the DB grows while being queried
        def appender():
            try:
                for i in range(0, N, 3):
                    # keyframes arrive slower than ticks in the live system: never run more than 6 rows ahead of the querier,
                    # so consecutive ticks are <= 9 rows apart and every 12-row revisit is seen by at least one tick
                    while i - last_tick[0] > 6 and not errors:
                        time.sleep(0)
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
                        # whatever prefix this tick saw, its answer must equal the oracle's for that prefix
                        if r.status == capi.CHIP_TICK_SCANNED and l % 7 == 0:
                            sc, ix = oracle_lib.scan_topk(db, l - 50, db[[l - 1, l - 2, l - 3]], 1)
                            assert list(r.argmax) == list(ix[:, 0]) and list(r.maxv) == list(sc[:, 0])
                    if done and chip.size() - last < 3:
                        break
            except Exception as e:  # pragma: no cover
                errors.append(e)

        def pnp_caller():
            try:
                while not stop.is_set():
                    p = capi.default_ransac_params(); p.seed = 9
                    g = chip.pnp_ransac(X, uv, p)
                    assert np.array_equal(g["T"], want_pnp["T"]) and np.array_equal(g["mask"], want_pnp["mask"])
                    pnp_runs[0] += 1
            except Exception as e:  # pragma: no cover
                errors.append(e)

        ts = [threading.Thread(target=f) for f in (appender, querier, pnp_caller)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errors, errors
        assert chip.size() == N and pnp_runs[0] > 0
        # every found loop is a correct statement about the DB prefix it was computed on
        assert len(found) > 0
        for cur, prev, score in found:
            assert score == oracle_lib.dot_tree(db[cur], db[prev]) and score > 0.85
