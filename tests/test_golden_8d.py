"""SURVEY 8d-conformant data: unit-L2 float32 rows of default_rng(20190412).standard_normal, ordinary revisits at
cosine 0.95 / 0.90 / 0.80, and three planted pairs whose score is EXACTLY (double)0.85f + 1 ulp / (double)0.85f / (double)0.85f - 1 ulp
in every summation order.  The strict accept rule of Cerebro.cpp:1056 (`u_max > THRESH`, THRESH = (double)(float)0.85, :913) fires on the
first and only on the first.  CPU: both oracle orders reproduce the committed ticks; GPU: the tick through the C ABI does, bit for bit."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib

GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))
import make_golden_8d as G  # noqa: E402


def fixture():
    g = json.loads((GOLD / "dot_scan_8d.json").read_text())
    db = G.build()
    if hashlib.sha256(db.tobytes()).hexdigest() != g["rows_sha256"]:
        pytest.skip(f"numpy {np.__version__} draws another standard_normal stream than numpy {g['numpy']} (fixture generated with the latter)")
    return g, db


def test_rows_are_unit_norm_and_thresholds_straddle():
    g, db = fixture()
    assert db.shape == (g["N"], g["D"]) and db.dtype == np.float32
    assert np.abs(np.linalg.norm(db.astype(np.float64), axis=1) - 1.0).max() < 3e-7
    th = float(np.float64(np.float32(0.85)))
    assert th.hex() == g["thresh_hex"] == "0x1.b333340000000p-1"
    by_l = {c["l"]: c for c in g["cases"]}
    t = g["straddling_ticks"]
    assert by_l[t["plus_1ulp"]]["maxv_hex"][0] == float(np.nextafter(th, 2.0)).hex() and by_l[t["plus_1ulp"]]["found"] == 1
    assert by_l[t["exact"]]["maxv_hex"][0] == th.hex() and by_l[t["exact"]]["found"] == 0                 # strict '>' (:1056)
    assert by_l[t["minus_1ulp"]]["maxv_hex"][0] == float(np.nextafter(th, 0.0)).hex() and by_l[t["minus_1ulp"]]["found"] == 0


def test_both_summation_orders_reproduce_the_fixture():
    g, db = fixture()
    for c in g["cases"]:
        tree, eig = oracle_lib.loop_tick_order(db, c["l"], 0), oracle_lib.loop_tick_order(db, c["l"], 1)
        for r, key in ((tree, "maxv_hex"), (eig, "maxv_hex_eigen_order")):
            assert (r["found"], r["idx_prev"], r["argmax"]) == (c["found"], c["idx_prev"], c["argmax"]), (c["l"], r)
            assert [float(x).hex() for x in r["maxv"]] == c[key]
        if c["l"] in g["straddling_ticks"].values():
            assert c["maxv_hex"][0] == c["maxv_hex_eigen_order"][0]          # one exact product + one exact term: no order can move it


@pytest.mark.gpu
def test_gpu_tick_on_8d_data_bit_exact():
    from cerebro_amd import capi
    g, db = fixture()
    with capi.Chip(g["D"]) as chip:
        chip.append_f64(db.astype(np.float64))                # the .srv wire type; narrowed on the device, verified lossless
        assert chip.info()["storage_bytes"] == 4
        for c in g["cases"]:
            chip.loop_reset()
            r = chip.loop_tick(c["l"]).as_dict()
            assert (r["found"], r["idx_prev"], r["argmax"]) == (c["found"], c["idx_prev"], c["argmax"]), (c["l"], r)
            assert [float(x).hex() for x in r["maxv"]] == c["maxv_hex"]
        # the same through the pipelined entry points and the top-k query
        ls = [c["l"] for c in g["cases"]]
        for s_, l in enumerate(ls):
            chip.loop_reset()
            chip.loop_tick_enqueue(l, s_)
        for s_, c in enumerate(g["cases"]):
            r = chip.loop_tick_collect(s_).as_dict()
            assert (r["found"], r["argmax"]) == (c["found"], c["argmax"])
        l = g["straddling_ticks"]["plus_1ulp"]
        sc, ix = chip.query_rows(l - 50, [l - 1], 4)
        want = oracle_lib.scan_topk(db, l - 50, db[[l - 1]], 4)
        assert np.array_equal(ix, want[1]) and np.array_equal(sc.view(np.uint64), want[0].view(np.uint64))
