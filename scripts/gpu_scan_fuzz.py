"""Fuzz (also run, smaller, by tests/test_fuzz_gpu.py) of the scan / top-k path vs the oracle: random D, N, prefix k, nq, topk, value distributions with many exact
ties (small-integer descriptors, duplicated rows), segment-crossing sizes, the many-query MFMA mode, and sharded layouts."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as O
from cerebro_amd import capi

def run(n=120, seed=11):
    rng = np.random.default_rng(seed)
    bad = []
    for it in range(n):
        D = int(rng.choice([4, 8, 60, 252, 256, 260, 1000, 1024, 2048, 4096, 4100, 8192]))
        N = int(rng.integers(1, 3000)) if D <= 4096 else int(rng.integers(1, 600))
        kind = it % 4
        if kind == 0: db = rng.standard_normal((N, D)).astype(np.float32)
        elif kind == 1: db = rng.integers(-2, 3, (N, D)).astype(np.float32)                  # many exact score ties
        elif kind == 2:
            db = rng.standard_normal((N, D)).astype(np.float32); db[rng.integers(0, N, N // 2)] = db[rng.integers(0, N)]   # duplicated rows
        else: db = (rng.standard_normal((N, D)) * 10.0 ** rng.integers(-20, 15)).astype(np.float32)   # extreme magnitudes
        k = int(rng.integers(0, N + 1))
        nq = int(rng.integers(1, 5)); topk = int(rng.integers(1, 17))
        with capi.Chip(D) as chip:
            cut = int(rng.integers(0, N + 1))
            if cut: chip.append_f64(db[:cut].astype(np.float64))
            if cut < N: chip.append_f32(db[cut:])
            q = db[rng.integers(0, N, nq)] if kind != 3 else rng.standard_normal((nq, D)).astype(np.float32)
            gs, gi = chip.query_vectors(k, q, topk)
            os_, oi = O.scan_topk(db, k, q, topk)
            ok = np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint64), os_.view(np.uint64))
            rows = rng.integers(0, N, nq)
            gs2, gi2 = chip.query_rows(k, rows, topk)
            os2, oi2 = O.scan_topk(db, k, db[rows], topk)
            ok = ok and np.array_equal(gi2, oi2) and np.array_equal(gs2.view(np.uint64), os2.view(np.uint64))
            if D % 32 == 0 and N >= 2 and it % 3 == 0:
                Q = int(rng.integers(1, 70))
                qq = db[rng.integers(0, N, Q)]
                bs, bi = chip.query_batch(k, qq, min(topk, 16))
                fs, fi = O.scan_topk_fmaf(db, k, qq, min(topk, 16))
                ok = ok and np.array_equal(bi, fi) and np.array_equal(bs.astype(np.float64).view(np.uint64), fs.view(np.uint64))
        if not ok:
            bad.append((it, D, N, k, nq, topk, kind))
    return bad


if __name__ == "__main__":
    t0 = time.time()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    bad = run(n)
    print(f"scan fuzz: {len(bad)} mismatches in {n} cases, {time.time()-t0:.1f} s")
    for b in bad[:10]: print(b)

