"""world_size-2 (and 3, and 8 = BASELINE config 4) CPU test of the sharded tick plumbing over the gloo backend: row->rank map, per-shard lists
with GLOBAL indices, all-gather layout, merge order (score desc, index desc) and the accept rule -- against the
unsharded oracle.  The device calls (chip_scan_local / chip_merge_decide) are played by an oracle-backed stand-in,
which is legitimate here because this test covers the HOST orchestration; the kernels themselves are covered by
tests/test_scan_gpu.py::test_sharded_scan_matches_single on the GPU."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
import scenarios
from cerebro_amd import capi, sharded


def test_row_map_matches_c_side():
    for G in (1, 2, 3, 8):
        for k in (0, 1, 5, 17, 100, 1001):
            counts = [sharded.local_count(k, r, G) for r in range(G)]
            assert sum(counts) == k and max(counts) - min(counts) <= 1
            for r in range(G):
                assert counts[r] == len([i for i in range(k) if sharded.owner_of(i, G) == r])
        for i in (0, 1, 7, 12345):
            assert sharded.global_index(sharded.local_index(i, G), sharded.owner_of(i, G), G) == i


class OracleShard:
    """Stand-in for a sharded chip_ctx: same scan_local / merge_decide contract, computed by the CPU oracle."""

    def __init__(self, db, rank, world):
        self.db, self.rank, self.world = db, rank, world
        self.rows = np.arange(rank, db.shape[0], world)
        self.last_l = 0
        self.p = oracle_lib.default_params()

    def set_stream(self, s):
        pass

    def scan_local(self, l, out_ptr, topk, params=None):
        p = self.p
        if l - self.last_l < p.min_new:
            return capi.CHIP_TICK_SKIPPED
        k = l - p.lag
        self.last_l = l
        if not k > p.min_k:
            return capi.CHIP_TICK_TOO_SHORT
        mine = self.rows[self.rows < k]
        sc, ix = oracle_lib.scan_topk(self.db[mine], len(mine), self.db[[l - 1, l - 2, l - 3]], topk)
        gix = np.where(ix >= 0, ix * self.world + self.rank, -1)            # local -> global index
        buf = np.empty((3, topk, 2), dtype=np.float64)
        buf[:, :, 0] = sc
        buf[:, :, 1] = gix.astype(np.int64).view(np.float64)
        C.memmove(out_ptr, buf.ctypes.data, buf.nbytes)
        return capi.CHIP_TICK_SCANNED

    def merge_decide(self, l, gathered_ptr, n_lists, topk, params=None):
        raw = np.empty((n_lists, 3, topk, 2), dtype=np.float64)
        C.memmove(raw.ctypes.data, gathered_ptr, raw.nbytes)
        sc, ix = raw[..., 0], raw[..., 1].copy().view(np.int64)
        r = capi.TickResult()
        r.status, r.idx_curr, r.idx_prev = capi.CHIP_TICK_SCANNED, -1, -1
        for q in range(3):
            cand = [(sc[g, q, j], ix[g, q, j]) for g in range(n_lists) for j in range(topk) if ix[g, q, j] >= 0]
            best = max(cand, key=lambda t: (t[0], t[1]))                     # (score desc, index desc)
            r.maxv[q], r.argmax[q] = best[0], int(best[1])
        p = self.p
        if abs(r.argmax[0] - r.argmax[1]) < p.locality and abs(r.argmax[0] - r.argmax[2]) < p.locality and r.maxv[0] > p.thresh:
            r.found, r.idx_curr, r.idx_prev, r.score = 1, l - 1, r.argmax[0], r.maxv[0]
        return r

    def merge_decide_enqueue(self, l, gathered_ptr, n_lists, slot, topk, params=None):
        self.slots = getattr(self, "slots", {})
        self.slots[slot] = self.merge_decide(l, gathered_ptr, n_lists, topk, params)   # the stand-in is synchronous

    def loop_tick_collect(self, slot):
        return self.slots.pop(slot)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, N = 256, 700
        plants, loops, ties = scenarios.loop_plants(N, 4, seed=5)
        db = scenarios.build_db(55, N, D, plants)
        det = sharded.ShardedLoopDetector(OracleShard(db, rank, world), topk=8, device="cpu")
        orc = oracle_lib.LoopOracle(db)
        n_found = 0
        for l in [3, 30, 57] + scenarios.default_schedule(N)[2:]:
            o = orc.tick(l)
            g = det.tick(l)
            assert g.status == o["status"], (l, g.status, o)
            if o["status"] == 2:
                assert list(g.argmax) == o["argmax"] and [float(x).hex() for x in g.maxv] == [float(x).hex() for x in o["maxv"]]
                assert (g.found, g.idx_curr, g.idx_prev) == (o["found"], o["idx_curr"], o["idx_prev"])
                n_found += g.found
        assert n_found >= len(loops)
        # pipelined form (double-buffered lists): same records, collected in enqueue order
        det2 = sharded.ShardedLoopDetector(OracleShard(db, rank, world), topk=8, device="cpu")
        orc2 = oracle_lib.LoopOracle(db)
        sched = scenarios.default_schedule(N)
        for base in range(0, len(sched), 7):
            chunk = sched[base:base + 7]
            sts = [det2.tick_enqueue(l, s) for s, l in enumerate(chunk)]
            for s, l in enumerate(chunk):
                o = orc2.tick(l)
                assert sts[s] == o["status"]
                if o["status"] == 2:
                    g = det2.collect(s)
                    assert (g.found, g.idx_curr, g.idx_prev, list(g.argmax)) == (o["found"], o["idx_curr"], o["idx_prev"], o["argmax"])
        ret[rank] = n_found
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_tick_over_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world and len(set(ret.values())) == 1
