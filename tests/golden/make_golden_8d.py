#!/usr/bin/env python3
"""SURVEY 8d-conformant descriptor data next to the integer-domain generator: unit-L2 float32 rows drawn from
numpy.random.default_rng(20190412).standard_normal (normalised in float32), ordinary revisits, and three planted query / row pairs whose
fp64 score is EXACTLY (double)0.85f + 1 ulp, (double)0.85f and (double)0.85f - 1 ulp -- in every summation order, because each is one
exact product plus one exactly representable term (q = (1, 2^-30, 0, ...), row = (0.85f, +-2^-23 | 0, sqrt(1 - 0.85f^2), 0, ...), each pair on its own four axes):
the accept rule of Cerebro.cpp:1056 (`u_max > THRESH`, THRESH = (double)(float)0.85, :913) must fire on the first and only on the first.
The expected ticks are produced by the CPU oracle in BOTH orders (fixed tree, Eigen 3.3 SSE2 GEMV) and committed; this script refuses
to overwrite the fixture.   python tests/golden/make_golden_8d.py"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))

SEED, N, D = 20190412, 2400, 4096
F085 = np.float32(0.85)
TICKS = {"plus_1ulp": 1000, "exact": 1500, "minus_1ulp": 2000}       # l of the three straddling ticks
ORDINARY = [(600, 200, 0.95), (1200, 420, 0.90), (1800, 700, 0.80)]   # (l, revisited row p, target cosine)


def build():
    rng = np.random.default_rng(SEED)
    db = rng.standard_normal((N, D)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True).astype(np.float32)          # float32 normalisation: unit L2 to float32 round-off
    third = np.float32(np.sqrt(np.float64(1.0) - np.float64(F085) ** 2))
    for t, (kind, l) in enumerate(TICKS.items()):
        p = l // 3                                                               # the planted row (and its two predecessors)
        e = 4 * t                                                                # each pair on its own axes: the special queries are orthogonal
        q = np.zeros(D, dtype=np.float32); q[e] = 1.0; q[e + 1] = np.float32(2.0 ** -30)
        r = np.zeros(D, dtype=np.float32); r[e] = F085; r[e + 2] = third
        r[e + 1] = {"plus_1ulp": np.float32(2.0 ** -23), "exact": np.float32(0.0), "minus_1ulp": np.float32(-(2.0 ** -23))}[kind]
        db[p] = r
        db[l - 1] = q
        db[l - 2] = db[p - 1]                                                    # the other two queries revisit the neighbours exactly
        db[l - 3] = db[p - 2]
    for l, p, cosv in ORDINARY:
        for j in range(3):
            n = rng.standard_normal(D).astype(np.float32)
            n -= np.float32(n @ db[p - j]) * db[p - j]
            n /= np.float32(np.linalg.norm(n))
            v = np.float32(cosv) * db[p - j] + np.float32(np.sqrt(1.0 - cosv * cosv)) * n
            db[l - 1 - j] = v / np.float32(np.linalg.norm(v))
    return np.ascontiguousarray(db)


def main():
    out = HERE / "dot_scan_8d.json"
    if out.exists() and "--force" not in sys.argv:
        raise SystemExit(f"{out} exists: fixtures are frozen (use a NEW file for new cases)")
    import oracle_lib
    db = build()
    norms = np.linalg.norm(db.astype(np.float64), axis=1)
    assert np.abs(norms - 1.0).max() < 3e-7
    thresh = float(np.float64(F085))
    ticks = sorted(list(TICKS.values()) + [l for l, _, _ in ORDINARY] + [300, 900, 2400])
    cases = []
    for l in ticks:
        per_order = []
        for order in (0, 1):                                   # 0 = the device's fixed tree, 1 = Eigen 3.3 SSE2 GEMV order
            st = oracle_lib.LoopOracle(db)
            st.state.last_l = 0
            per_order.append(oracle_lib.loop_tick_order(db, l, order) if hasattr(oracle_lib, "loop_tick_order") else st.tick(l))
        a, b = per_order
        assert a["found"] == b["found"] and a["argmax"] == b["argmax"], (l, a, b)
        cases.append(dict(l=l, found=a["found"], idx_prev=a["idx_prev"], argmax=a["argmax"], maxv_hex=[float(x).hex() for x in a["maxv"]],
                          maxv_hex_eigen_order=[float(x).hex() for x in b["maxv"]]))
    by_l = {c["l"]: c for c in cases}
    up = float(np.nextafter(thresh, 2.0)); dn = float(np.nextafter(thresh, 0.0))
    assert by_l[TICKS["plus_1ulp"]]["maxv_hex"][0] == up.hex() and by_l[TICKS["plus_1ulp"]]["found"] == 1
    assert by_l[TICKS["exact"]]["maxv_hex"][0] == thresh.hex() and by_l[TICKS["exact"]]["found"] == 0
    assert by_l[TICKS["minus_1ulp"]]["maxv_hex"][0] == dn.hex() and by_l[TICKS["minus_1ulp"]]["found"] == 0
    doc = {"what": "SURVEY 8d generator: unit-L2 float32 rows of default_rng(20190412).standard_normal + planted scores at (double)0.85f and +-1 ulp",
           "generator": "tests/golden/make_golden_8d.py (build())", "seed": SEED, "N": N, "D": D, "numpy": np.__version__,
           "rows_sha256": hashlib.sha256(db.tobytes()).hexdigest(), "thresh_hex": thresh.hex(),
           "straddling_ticks": TICKS, "ordinary_revisits": ORDINARY, "max_norm_deviation": float(np.abs(norms - 1.0).max()), "cases": cases}
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(f"wrote {out}: {len(cases)} ticks, found at {[c['l'] for c in cases if c['found']]}")


if __name__ == "__main__":
    main()
