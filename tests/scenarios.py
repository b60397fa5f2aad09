"""Deterministic synthetic scenarios shared by the CPU and GPU tests (inputs are regenerated from the
integer-domain generator spec, so only seeds + expected outputs are committed under tests/golden/)."""
from __future__ import annotations

import numpy as np

import oracle_lib


def loop_plants(N: int, n_loops: int, seed: int, lag: int = 50, with_ties: bool = True):
    """Plant `n_loops` revisits: rows (q-2,q-1,q) become noisy copies of (p-2,p-1,p), p + lag + 60 < q.
    Optionally add exact-duplicate rows so that the last-index tie rule (Cerebro.cpp:1038-1043) is exercised:
    rows t1 < t2 are bit-identical copies of row s (< t1), and the query row's source is s."""
    rng = np.random.default_rng(seed)
    plants = {}
    loops = []
    # q must be of the form l-1 with l = 56 + 3*i (default tick schedule) so the tick queries exactly rows q,q-1,q-2
    cand_l = np.arange(56 + 3 * 40, N + 1, 3)
    used = set()
    tries = 0
    while len(loops) < n_loops and tries < 10000:
        tries += 1
        l = int(rng.choice(cand_l))
        q = l - 1
        p = int(rng.integers(10, q - lag - 60))
        rows = {q, q - 1, q - 2, p, p - 1, p - 2}
        if any(abs(r - u) < 6 for r in rows for u in used):
            continue
        used |= rows
        for j in range(3):
            plants[q - j] = (p - j, 1)
        loops.append((l, q, p))
    ties = []
    if with_ties and loops:
        # make the newest loop's target ambiguous: two later exact duplicates of its source row p
        l, q, p = loops[0]
        t1, t2 = p + 3, p + 5   # other planted rows are >= 6 away from p; |t2 - (p-2)| = 7 < 12 keeps locality
        plants[t1] = (p, 2)
        plants[t2] = (p, 2)
        ties.append((p, t1, t2))
    return sorted((d, s, k) for d, (s, k) in plants.items()), loops, ties


def build_db(seed: int, N: int, D: int, plants):
    return oracle_lib.synth_rows(seed, range(N), D, plants)


def default_schedule(N: int):
    """SURVEY 8d tick schedule: l advances by exactly 3 per tick from 56 (first k > 5)."""
    return list(range(56, N + 1, 3))


_synth_topk_cache = {}


def cached_scan_topk_synth(seed: int, k: int, D: int, query_rows, K: int, plants, nthreads: int):
    """oracle_lib.scan_topk_synth with a per-process cache: the full-size (1M-row) CPU-oracle scan takes tens of seconds and
    several tests (unsharded, 8 sub-contexts on one device, real multi-GPU layouts) check the same tick against it."""
    key = (seed, k, D, tuple(int(r) for r in query_rows), K, tuple(tuple(int(x) for x in p) for p in plants))
    if key not in _synth_topk_cache:
        q = oracle_lib.synth_rows(seed, list(query_rows), D, plants)
        _synth_topk_cache[key] = oracle_lib.scan_topk_synth(seed, k, D, q, K, plants, nthreads=nthreads)
    return _synth_topk_cache[key]
