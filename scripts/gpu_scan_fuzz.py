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
        D = int(rng.choice([4, 8, 60, 252, 256, 260, 1000, 1024, 2048, 3072, 4096, 4100, 5120, 8192]))   # multiples of 1024: the asm load path
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


def run_ticks(n=12, seed=5):
    """Random tick schedules (skips, jumps, too-short prefixes), random accept-rule parameters, synchronous and pipelined
    ticks, and a random shard count emulated on one GPU -- every record against the oracle's."""
    import torch
    import scenarios
    rng = np.random.default_rng(seed)
    bad = []
    def same(g, o):
        g = g.as_dict()
        return all(g[k] == o[k] for k in ("status", "found", "idx_curr", "idx_prev", "argmax")) and \
            float(g["score"]).hex() == float(o["score"]).hex() and [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]
    for it in range(n):
        D = int(rng.choice([64, 256, 512, 1000])); N = int(rng.integers(300, 1200))
        plants, loops, ties = scenarios.loop_plants(N, 4, seed=100 + it, lag=int(rng.integers(10, 60)))
        db = scenarios.build_db(200 + it, N, D, plants)
        vals = dict(locality=int(rng.integers(1, 25)), thresh=float(rng.choice([0.5, 0.85000002384185791015625, 0.95])),
                    lag=int(rng.integers(5, 80)), min_new=int(rng.integers(1, 6)), min_k=int(rng.integers(0, 12)))
        ls, l = [], 0
        while True:
            l += int(rng.choice([1, 2, 3, 3, 3, 4, 7, 20]))
            if l > N: break
            ls.append(l)
        op = O.OrcDotParams(vals["locality"], vals["thresh"], vals["lag"], vals["min_new"], vals["min_k"])
        want = []
        orc = O.LoopOracle(db, op)
        for l in ls: want.append(orc.tick(l))
        gp = capi.default_dot_params()
        for k_, v_ in vals.items(): setattr(gp, k_, v_)
        with capi.Chip(D) as chip:
            chip.append_f32(db)
            got = [chip.loop_tick(l, gp) for l in ls]
            if not all(same(g, o) for g, o in zip(got, want)): bad.append(("sync", it, vals))
            chip.loop_reset()
            W = int(rng.integers(1, 9)); pend = []; got = []
            for i, l in enumerate(ls):
                if len(pend) == W: got.append(chip.loop_tick_collect(pend.pop(0)))
                chip.loop_tick_enqueue(l, i % W, gp); pend.append(i % W)
            while pend: got.append(chip.loop_tick_collect(pend.pop(0)))
            if not all(same(g, o) for g, o in zip(got, want)): bad.append(("pipelined", it, W, vals))
        G = int(rng.choice([2, 3, 5, 8])); K = int(rng.integers(1, 17))
        chips = [capi.Chip(D, shard_rank=r, shard_count=G) for r in range(G)]
        try:
            for c in chips: c.append_f32(db)
            bufs = torch.zeros((G, 3, K, 2), dtype=torch.float64, device="cuda")
            for l, o in zip(ls, want):
                st = [c.scan_local(l, bufs[r].data_ptr(), K, gp) for r, c in enumerate(chips)]
                if len(set(st)) != 1 or st[0] != o["status"]: bad.append(("shard-status", it, l)); break
                if st[0] != capi.CHIP_TICK_SCANNED: continue
                for c in chips: c.synchronize()
                if not all(same(c.merge_decide(l, bufs.data_ptr(), G, K, gp), o) for c in chips): bad.append(("sharded", it, G, K, l)); break
        finally:
            for c in chips: c.close()
    return bad


def run_f64_and_groups(n=40, seed=23):
    """Round-2 surface under the same randomised treatment: double-row DBs (genuinely float64 values, ties, duplicated rows,
    extreme magnitudes; chosen explicitly or by the automatic switch), groups of 1..8 sub-contexts on one device (float and
    double rows), score-vector export, tick schedules through the group entry points -- all bit for bit vs the oracle."""
    import scenarios
    rng = np.random.default_rng(seed)
    bad = []
    for it in range(n):
        f64 = it % 2 == 0
        G = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))                     # 0 = plain ctx
        D = int(rng.choice([4, 60, 252, 256, 512, 1000, 1024, 1536, 4096] + ([6824] if f64 else [3072, 8192])))   # whole 4-KiB batches (asm loads) and ragged rows
        N = int(rng.integers(60, 1500)) if D <= 4096 else int(rng.integers(60, 300))
        kind = it % 4
        if kind == 0: db = rng.standard_normal((N, D))
        elif kind == 1: db = rng.integers(-2, 3, (N, D)).astype(np.float64) * (1.0 + 2.0 ** -40 if f64 else 1.0)   # exact ties
        elif kind == 2:
            db = rng.standard_normal((N, D)); db[rng.integers(0, N, N // 2)] = db[rng.integers(0, N)]
        else: db = rng.standard_normal((N, D)) * 10.0 ** rng.integers(-20, 15)
        if not f64: db = db.astype(np.float32).astype(np.float64)
        elif kind != 1: db[0, 0] = 0.1                                   # make sure the first append is not float32-representable
        scan = (lambda k, q, K: O.scan_topk_f64(db, k, q, K)) if f64 else (lambda k, q, K: O.scan_topk(db.astype(np.float32), k, q, K))
        kw = dict(devices=[0] * G) if G else {}
        storage = None if (it % 3 == 0 and (not f64 or kind != 1)) else ("f64" if f64 else "f32")    # None: decided by the data
        k = int(rng.integers(0, N + 1)); nq = int(rng.integers(1, 4 if f64 and D * 8 * 4 > 160 * 1024 else 5)); topk = int(rng.integers(1, 17))
        with capi.Chip(D, storage=storage, **kw) as chip:
            cut = int(rng.integers(1, N + 1))
            chip.append_f64(db[:cut])
            if cut < N: chip.append_f64(db[cut:])
            ok = chip.info()["storage_bytes"] == (8 if f64 else 4) and chip.size() == N
            q = db[rng.integers(0, N, nq)]
            gs, gi = chip.query_vectors_f64(k, q, topk)
            os_, oi = scan(k, q if f64 else q.astype(np.float32), topk)
            ok = ok and np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint64), os_.view(np.uint64))
            lo = max(0, N - 4000) if G > 1 else 0                        # sharded layouts: query rows come from the replicated ring
            rows = rng.integers(lo, N, nq)
            gs2, gi2 = chip.query_rows(k, rows, topk)
            os2, oi2 = scan(k, db[rows] if f64 else db[rows].astype(np.float32), topk)
            ok = ok and np.array_equal(gi2, oi2) and np.array_equal(gs2.view(np.uint64), os2.view(np.uint64))
            u = chip.query_scores(k, int(rows[0]))
            want_u = O.scores(db if f64 else db.astype(np.float32), k, db[rows[0]] if f64 else db[rows[0]].astype(np.float32))
            ok = ok and np.array_equal(u.view(np.uint64), want_u.view(np.uint64))
            back = chip.read_rows_f64([0, N - 1])
            ok = ok and back.tobytes() == db[[0, N - 1]].tobytes()
        if not ok:
            bad.append(("scan", it, D, N, k, nq, topk, kind, f64, G, storage))
    for it in range(max(2, n // 8)):                                    # tick schedules through groups, float and double rows
        D = int(rng.choice([64, 256, 512])); N = int(rng.integers(300, 1000)); G = int(rng.choice([1, 2, 3, 8])); f64 = it % 2 == 1
        plants, loops, ties = scenarios.loop_plants(N, 4, seed=300 + it)
        db32 = scenarios.build_db(400 + it, N, D, plants)
        db = db32.astype(np.float64)
        if f64:
            db = db * (1.0 + 2.0 ** -40 * rng.standard_normal((N, D)))
            for d_, s_, k_ in plants:
                if k_ == 2: db[d_] = db[s_]
        orc = O.LoopOracle64(db) if f64 else O.LoopOracle(db32)
        ls, l = [], 0
        while True:
            l += int(rng.choice([1, 2, 3, 3, 3, 4, 7, 20]))
            if l > N: break
            ls.append(l)
        want = [orc.tick(l) for l in ls]
        def same(g, o):
            g = g.as_dict()
            return all(g[k] == o[k] for k in ("status", "found", "idx_curr", "idx_prev", "argmax")) and \
                float(g["score"]).hex() == float(o["score"]).hex() and [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]
        with capi.Chip(D, devices=[0] * G) as chip:
            chip.append_f64(db)
            got = [chip.loop_tick(l) for l in ls]
            if not all(same(g, o) for g, o in zip(got, want)): bad.append(("group-sync", it, G, f64))
            chip.loop_reset()
            W = int(rng.integers(1, 20)); pend = []; got = []
            for i, l in enumerate(ls):
                if len(pend) == W: got.append(chip.loop_tick_collect(pend.pop(0)))
                chip.loop_tick_enqueue(l, i % W); pend.append(i % W)
            while pend: got.append(chip.loop_tick_collect(pend.pop(0)))
            if not all(same(g, o) for g, o in zip(got, want)): bad.append(("group-pipelined", it, G, W, f64))
    return bad


if __name__ == "__main__":
    t0 = time.time()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    bad = run(n)
    print(f"scan fuzz: {len(bad)} mismatches in {n} cases, {time.time()-t0:.1f} s")
    for b in bad[:10]: print(b)
    sys.path.insert(0, 'tests')
    bt = run_ticks(12)
    print(f"tick fuzz: {len(bt)} mismatches in 12 schedules")
    for b in bt[:10]: print(b)
    bf = run_f64_and_groups(80)
    print(f"f64 / group fuzz: {len(bf)} mismatches in 80 + 10 cases")
    for b in bf[:10]: print(b)

