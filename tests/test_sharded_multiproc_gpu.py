"""The sharded tick with REAL processes and the real HIP kernels: `world` processes share the one GPU of the test box, each with
its own sharded ctx (rank r holds rows i % world == r), exchanging the per-shard lists through torch.distributed.  RCCL refuses
two ranks on one device, so the transport here is gloo on device tensors; what is exercised is everything else of the N-GPU
path: per-process contexts and streams, scan_local -> exchange -> merge ordering, the pipelined form, and that every rank
reaches the oracle's decision."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
import scenarios

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cerebro_amd import capi, sharded
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, N = 512, 1500
        plants, loops, ties = scenarios.loop_plants(N, 5, seed=9)
        db = scenarios.build_db(91, N, D, plants)
        sched = [3, 30, 57] + scenarios.default_schedule(N)[2:]
        with capi.Chip(D, shard_rank=rank, shard_count=world) as chip:
            chip.append_f32(db[:400]); chip.append_f64(db[400:].astype(np.float64))
            det = sharded.ShardedLoopDetector(chip, topk=8, device=torch.device("cuda", 0))
            orc = oracle_lib.LoopOracle(db)
            n_found = 0
            for l in sched:
                o = orc.tick(l)
                g = det.tick(l)
                assert g.status == o["status"], (l, g.status, o)
                if o["status"] == capi.CHIP_TICK_SCANNED:
                    assert list(g.argmax) == o["argmax"] and [float(x).hex() for x in g.maxv] == [float(x).hex() for x in o["maxv"]]
                    assert (g.found, g.idx_curr, g.idx_prev) == (o["found"], o["idx_curr"], o["idx_prev"])
                    n_found += g.found
            assert n_found >= len(loops)
            # pipelined: several ticks in flight, collected in order
            chip.loop_reset()
            orc2 = oracle_lib.LoopOracle(db)
            W = 6
            for base in range(3, len(sched), W):
                chunk = sched[base:base + W]
                sts = [det.tick_enqueue(l, s) for s, l in enumerate(chunk)]
                for s, l in enumerate(chunk):
                    o = orc2.tick(l)
                    assert sts[s] == o["status"]
                    if o["status"] == capi.CHIP_TICK_SCANNED:
                        g = det.collect(s)
                        assert (g.found, g.idx_curr, g.idx_prev, list(g.argmax)) == (o["found"], o["idx_curr"], o["idx_prev"], o["argmax"])
            det.close()
        ret[rank] = n_found
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_tick_real_processes_one_gpu(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world and len(set(ret.values())) == 1
