"""The in-library one-process-per-GPU exchange (chip_comm_init_rank + ncclAllGather / ncclBroadcast, BASELINE config 4's launch
shape, the thread wiring of cerebro_node.cpp:487,499,509) with MORE THAN ONE RANK on a 1-GPU box.

Real RCCL refuses two ranks on one device, so until round 5 this code had only ever run at world size 1.  Here N real processes
share device 0 and libcerebro_hip.so loads tests/fakerccl/_build/libfakerccl.so (a shared-memory stand-in for the eight librccl
symbols it binds -- test infrastructure, selected through the library's own CHIP_RCCL_LIBRARY override) instead of librccl:
every collective call of chip_multi.hip is sequenced exactly as on an 8-GPU node -- local scan -> local merge -> all-gather ->
merge + decision on every rank; owner fetch (broadcast) of old query rows; the agreement round + large all-gather of the
many-query mode; the failure mark of a shard that cannot take part; a rank whose allocations fail.  Every rank checks every
result bit for bit against the UNSHARDED CPU oracle.  No performance is measured through the stub."""
import json
import os
import socket
import subprocess
import time
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import HOOKS_ENV   # fault injection exists in the TEST build of the library only (cerebro_amd/lib/hooks/)

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
FAKE = ROOT / "tests" / "fakerccl" / "_build" / "libfakerccl.so"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _worker(rank, world, uid_path, ret, mode):
    import time
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    import scenarios
    from cerebro_amd import capi
    D, N = 1024, 1500
    plants, loops, ties = scenarios.loop_plants(N, 5, seed=21 + world)
    db = scenarios.build_db(733 + world, N, D, plants)
    with capi.Chip(D, device=0, shard_rank=rank, shard_count=world) as chip:       # every rank on device 0
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(capi.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        chip.comm_init_rank(open(uid_path, "rb").read(), world, rank)
        info = chip.info()
        assert info["exchange"] == capi.CHIP_EXCHANGE_RCCL and info["comm_ranks"] == world, info
        assert info["rows_local"] == 0
        chip.append_f32(db[:700])
        chip.append_f64(db[700:].astype(np.float64))
        assert chip.info()["rows_local"] == len(range(rank, N, world))               # this rank stores only its residue class
        orc = oracle_lib.LoopOracle(db)
        sched = scenarios.default_schedule(N)
        n_found = n_failed = 0

        def same(g, o, l):
            for key in ("status", "found", "idx_curr", "idx_prev", "argmax"):
                assert g[key] == o[key], (rank, l, key, g, o)
            assert [float(x).hex() for x in g["maxv"]] == [float(x).hex() for x in o["maxv"]]

        # ---- synchronous ticks (collective: every rank makes the same calls in the same order)
        for l in sched:
            before = chip.last_l()
            try:
                g = chip.loop_tick(l).as_dict()
            except capi.ChipError as e:       # mode "fail": rank 1 sends the failure mark on every 5th collective call -- EVERY rank sees it
                assert mode == "fail" and e.status == capi.CHIP_ERR_SHARD_FAILED, (rank, l, e)
                assert chip.last_l() == before
                n_failed += 1
                g = chip.loop_tick(l).as_dict()      # retry: the exchange is still in step
            same(g, orc.tick(l), l)
            n_found += g["found"]
        assert n_found >= len(loops)
        if mode == "fail":
            assert n_failed >= 10
            ret[rank] = (n_found, n_failed)
            return
        # ---- pipelined ticks: 4 in flight, collected in order
        chip.loop_reset()
        orc2 = oracle_lib.LoopOracle(db)
        for base in range(0, 40, 4):
            chunk = sched[base:base + 4]
            for s_, l in enumerate(chunk):
                chip.loop_tick_enqueue(l, s_)
            for s_, l in enumerate(chunk):
                same(chip.loop_tick_collect(s_).as_dict(), orc2.tick(l), l)
        # ---- top-k queries: newest rows (ring) and OLD rows (fetched from their owners by broadcast), every K
        for rows, K in (([N - 1, N - 2, N - 3], 8), ([5, 6, 7, 8], 16), ([loops[0][1]], 1), ([N - 1, 3, 700, 701], 3)):
            got, want = chip.query_rows(N - 50, rows, K), oracle_lib.scan_topk(db, N - 50, db[rows], K)
            assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0])), (rank, rows, K)
        # external query vectors, a prefix that cuts through the residue classes
        qv = db[[11, 400]]
        got, want = chip.query_vectors(777, qv, 5), oracle_lib.scan_topk(db, 777, qv, 5)
        assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        # ---- the score vector of one (old) query row: this rank fills the entries of the rows it owns
        u = chip.query_scores(N - 50, 123)
        full = oracle_lib.scores(db, N - 50, db[123])
        mine = np.arange(rank, N - 50, world)
        assert np.array_equal(bits(u[mine]), bits(full[mine]))
        # ---- many-query (MFMA) mode: agreement round + all-gather of Q x topk entries per rank + merge on every rank
        for Q, K in ((70, 8), (130, 16)):
            q = db[(np.arange(Q) * 13) % N]
            sc, ix = chip.query_batch(N - 50, q, K)
            wsc, wix = oracle_lib.scan_topk_fmaf(db, N - 50, q, K)
            assert np.array_equal(ix, wix) and np.array_equal(sc.view(np.uint32), wsc.astype(np.float32).view(np.uint32)), (rank, Q, K)
        if mode == "oom":     # CHIP_TEST_BATCH_OOM=1: rank 1's allocations "fail" -> nobody posts the large all-gather
            raise AssertionError("mode oom must not reach a successful many-query call")
        ret[rank] = (n_found, 0)


def _worker_oom(rank, world, uid_path, ret):
    """Rank 1 of 2 cannot allocate its many-query buffers: it returns CHIP_ERR_OOM, rank 0 CHIP_ERR_SHARD_FAILED, and the NEXT
    collective calls (a tick, a top-k query) pair up correctly -- the communicator never went out of step."""
    import time
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    import scenarios
    from cerebro_amd import capi
    D, N = 512, 900
    plants, loops, ties = scenarios.loop_plants(N, 3, seed=5)
    db = scenarios.build_db(99, N, D, plants)
    with capi.Chip(D, device=0, shard_rank=rank, shard_count=world) as chip:
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(capi.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        chip.comm_init_rank(open(uid_path, "rb").read(), world, rank)
        chip.append_f32(db)
        q = db[:64]
        for attempt in range(2):
            with pytest.raises(capi.ChipError) as ei:
                chip.query_batch(N - 50, q, 8)
            assert ei.value.status == (capi.CHIP_ERR_OOM if rank == 1 else capi.CHIP_ERR_SHARD_FAILED), (rank, ei.value.status)
            l = N - attempt
            chip.loop_reset()
            g, o = chip.loop_tick(l).as_dict(), oracle_lib.LoopOracle(db).tick(l)
            assert g["status"] == o["status"] and g["argmax"] == o["argmax"], (rank, g, o)
            got, want = chip.query_rows(N - 50, [N - 1, 2], 4), oracle_lib.scan_topk(db, N - 50, db[[N - 1, 2]], 4)
            assert np.array_equal(got[1], want[1]) and np.array_equal(bits(got[0]), bits(want[0]))
        ret[rank] = (1, 0)


def _spawn(fn, world, tmp_path, args, env):
    import torch.multiprocessing as mp
    assert FAKE.exists(), "tests/fakerccl/_build/libfakerccl.so is not built (make fakerccl)"
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)                      # spawned children inherit it; the library reads CHIP_RCCL_LIBRARY at its first RCCL use
    try:
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(fn, args=(world, str(tmp_path / "uid.bin"), ret) + args, nprocs=world, join=True)
        return dict(ret)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


BASE_ENV = {"CHIP_RCCL_LIBRARY": str(FAKE), "FAKERCCL_TIMEOUT_MS": "60000", "CHIP_COMM_INIT_TIMEOUT_MS": "90000"}


@pytest.mark.parametrize("world,enqueued", [(2, 0), (4, 0), (2, 1), (4, 1)])
def test_exchange_with_real_processes_on_one_device(world, enqueued, tmp_path):
    """enqueued = 1 (round 6): the stand-in ENQUEUES the small collectives on the caller's stream (host function between two async
    copies) instead of executing them at call time -- the in-stream form the real collective has; same parity bar."""
    ret = _spawn(_worker, world, tmp_path, ("ok",), dict(BASE_ENV, FAKERCCL_ASYNC=str(enqueued)))
    assert len(ret) == world and len(set(ret.values())) == 1 and next(iter(ret.values()))[0] > 0


def test_failed_shard_mark_across_processes(tmp_path):
    """Rank 1 of 3 cannot take part in every 5th collective call: it sends the marked neutral list, the merge of EVERY rank reports
    the mark (CHIP_ERR_SHARD_FAILED everywhere, last_l unchanged), the retry succeeds -- across real process boundaries."""
    ret = _spawn(_worker, 3, tmp_path, ("fail",), dict(BASE_ENV, CHIP_TEST_FAIL_SHARD="1:5", **HOOKS_ENV))
    assert len(ret) == 3 and len(set(ret.values())) == 1 and next(iter(ret.values()))[1] >= 10


def test_rank_out_of_memory_in_many_query_call(tmp_path):
    ret = _spawn(_worker_oom, 2, tmp_path, (), dict(BASE_ENV, CHIP_TEST_BATCH_OOM="1", **HOOKS_ENV))
    assert len(ret) == 2


def _bench_torchrun(n, env_extra, rows="60000", steps="24"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", **BASE_ENV, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(n), "--steps", steps, "--warmup", "4", "--rows", rows]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), took


@pytest.mark.parametrize("n", [4, 8])
def test_bench_under_torchrun_reports_rccl_ranks(n):
    """bench.py launched the way the driver launches N > 1 (N = 8: the driver's scaling run), ranks sharing device 0, the library's
    exchange over the stub: the JSON line says the in-library collective spanned N ranks (config.rccl_ranks == N, no fallback), within
    a bounded time."""
    j, took = _bench_torchrun(n, {})
    assert j["n_gpus"] == n and j["value"] > 0 and j["scaling"] == "strong"
    assert j["config"]["rccl_ranks"] == n and not j["config"]["exchange_fallback"], j["config"]
    assert "in-library RCCL" in j["config"]["exchange"]
    assert took < 300, took


def test_bench_one_process_eight_sub_contexts_over_the_stand_in():
    """The OTHER layout of the driver's 8-GPU run (python bench.py --gpus 8 without torchrun = one process, chip_create_multi ->
    ncclCommInitAll): eight sub-contexts of device 0, the stand-in's ncclCommInitAll (real RCCL refuses repeated devices; the TEST
    build of the library lets the group ask for the RCCL transport anyway): rccl_ranks == 8, a JSON line, bounded time."""
    env = dict(os.environ, **BASE_ENV, **HOOKS_ENV, CHIP_TEST_RCCL_SAME_DEVICE="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--same-device", "--steps", "24", "--warmup", "4", "--rows", "60000",
                        "--cpu-budget", "0", "--no-pnp", "--no-batch"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["config"]["process_layout"] == "one process"
    assert j["config"]["rccl_ranks"] == 8 and "in-library RCCL (ncclCommInitAll" in j["config"]["exchange"], j["config"]
    assert took < 300, took


def test_enqueued_all_gather_overlaps_the_next_scan():
    """with the collective ENQUEUED in-stream, all-gather(i) sits on the ctx stream while scan(i+1) runs on
    the scan streams -- the overlap the >= 6x prediction rests on.  Timed on one GPU at 4 ranks x 125k rows of 4096-D (each rank's scan
    ~0.3 ms when alone; here they share one device, so the absolute numbers say nothing about xGMI): the pipelined tick loop must not be
    slower with enqueued collectives than with collectives executed at call time (which serialise host and stream per tick), and both
    runs fire the planted revisits on every rank.  Numbers -> gpurun_out/r06/fakerccl_overlap.json."""
    res = {}
    for mode in ("0", "1", "0", "1"):
        j, took = _bench_torchrun(4, {"FAKERCCL_ASYNC": mode}, rows="500000", steps="60")
        assert j["config"]["rccl_ranks"] == 4 and not j["config"]["exchange_fallback"]
        res.setdefault(mode, []).append(j["ms_per_step"])
    sync_ms, async_ms = min(res["0"]), min(res["1"])
    rep = {"ranks": 4, "rows": 500000, "ms_per_step_collective_at_call_time": res["0"], "ms_per_step_collective_enqueued": res["1"],
           "note": "4 processes on ONE device over tests/fakerccl (shared memory): sequencing and overlap, not xGMI"}
    out = ROOT / "gpurun_out" / "r06"
    try:
        out.mkdir(parents=True, exist_ok=True)
        (out / "fakerccl_overlap.json").write_text(json.dumps(rep, indent=1))
    except OSError:
        pass
    print(json.dumps(rep))
    assert async_ms <= 1.10 * sync_ms, rep


def _worker_1m(rank, world, uid_path, ret, want_path):
    import time
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from cerebro_amd import capi
    w = np.load(want_path)
    wsc, wix = w["sc"], w["ix"]
    D, N, seed = 4096, 1_000_053, 20190412
    l = N
    q, p = l - 1, 777_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2), (123_456, p - 1, 2)]
    with capi.Chip(D, capacity_hint=N, device=0, shard_rank=rank, shard_count=world) as chip:
        if rank == 0:
            with open(uid_path + ".tmp", "wb") as f:
                f.write(capi.comm_unique_id())
            os.replace(uid_path + ".tmp", uid_path)
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        chip.comm_init_rank(open(uid_path, "rb").read(), world, rank)
        assert chip.info()["comm_ranks"] == world
        chip.append_synthetic(N, seed, plants)
        assert chip.size() == N and chip.info()["rows_local"] == len(range(rank, N, world))
        r = chip.loop_tick(l)
        assert r.status == capi.CHIP_TICK_SCANNED and r.found == 1 and r.idx_curr == q and r.idx_prev == p + 4
        assert list(r.argmax) == list(wix[:, 0]) == [p + 4, p - 1, p - 2]
        assert [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        got = chip.query_rows(l - 50, [l - 1, l - 2, l - 3], 8)
        assert np.array_equal(got[1], wix) and np.array_equal(bits(got[0]), bits(wsc))
        ret[rank] = (1, 0)


def test_config4_shape_one_process_per_rank_at_full_size(tmp_path):
    """BASELINE config 4 in its own launch shape AND at its own size: 4096-D x 1 000 053 rows row-sharded over EIGHT processes (here
    all on device 0, 125k rows = 2 GB each), the per-shard top-k lists all-gathered inside the library (over the RCCL stand-in), merge
    + decision on every rank: the tick and the top-8 lists are those of the full CPU-oracle scan, bit for bit, on every rank."""
    sys.path.insert(0, str(ROOT / "tests"))
    import scenarios
    D, N, seed = 4096, 1_000_053, 20190412
    l = N
    q, p = l - 1, 777_777
    plants = [(q - j, p - j, 1) for j in range(3)] + [(p + 4, p, 2), (123_456, p - 1, 2)]
    wsc, wix = scenarios.cached_scan_topk_synth(seed, l - 50, D, [l - 1, l - 2, l - 3], 8, plants, nthreads=min(os.cpu_count() or 1, 128))
    np.savez(tmp_path / "want.npz", sc=wsc, ix=wix)
    ret = _spawn(_worker_1m, 8, tmp_path, (str(tmp_path / "want.npz"),), BASE_ENV)
    assert len(ret) == 8


def test_missing_peer_at_the_rendezvous_is_an_error_not_a_hang(tmp_path, monkeypatch):
    """World 2, only rank 0 shows up: the communicator bootstrap cannot complete.  chip_comm_init_rank returns CHIP_ERR_COMM (the
    stand-in's rendezvous deadline here; the library's own CHIP_COMM_INIT_TIMEOUT_MS helper-thread deadline covers a bootstrap that
    never returns) -- through the real entry point, no fault-injection hook."""
    import time
    code = (
        "import sys, time\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from cerebro_amd import capi\n"
        "with capi.Chip(256, device=0, shard_rank=0, shard_count=2) as chip:\n"
        "    t0 = time.time()\n"
        "    try:\n"
        "        chip.comm_init_rank(capi.comm_unique_id(), 2, 0)\n"
        "        print('UNEXPECTED ok')\n"
        "    except capi.ChipError as e:\n"
        "        print('status', e.status, 'after', round(time.time() - t0, 1), 's', 'exchange', chip.info()['exchange'])\n")
    env = dict(os.environ, CHIP_RCCL_LIBRARY=str(FAKE), FAKERCCL_TIMEOUT_MS="1500", CHIP_COMM_INIT_TIMEOUT_MS="20000")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "status -11" in r.stdout and "exchange 0" in r.stdout, r.stdout + r.stderr[-500:]      # CHIP_ERR_COMM, no exchange attached
    assert time.time() - t0 < 60


def _worker_peer_dies(rank, world, uid_path, ret):
    import time
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import scenarios
    from cerebro_amd import capi
    D, N = 256, 600
    db = scenarios.build_db(3, N, D, [])
    chip = capi.Chip(D, device=0, shard_rank=rank, shard_count=world)
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(capi.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    t0 = time.time()
    while not os.path.exists(uid_path):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    chip.comm_init_rank(open(uid_path, "rb").read(), world, rank)
    chip.append_f32(db)
    for l in (300, 303, 306):
        assert chip.loop_tick(l).status == capi.CHIP_TICK_SCANNED
    if rank == 1:
        ret[rank] = "left"
        os._exit(0)                                   # the peer vanishes between two collective calls
    t1 = time.time()
    try:
        chip.loop_tick(309)
        ret[rank] = "UNEXPECTED ok"
    except capi.ChipError as e:
        ret[rank] = (e.status, round(time.time() - t1, 1), chip.last_comm_error())
    os._exit(0)                                       # (the communicator has a dead member: no orderly teardown)


def test_a_peer_that_dies_fails_the_next_collective_instead_of_hanging_it(tmp_path):
    """Rank 1 of 2 exits after three ticks.  Rank 0's next tick posts its all-gather, the peer never arrives: the call comes back with
    CHIP_ERR_COMM and chip_last_comm_error() says why -- bounded by the transport's deadline, not a hang."""
    import torch.multiprocessing as mp
    env = dict(BASE_ENV, FAKERCCL_TIMEOUT_MS="2000")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        mgr = mp.Manager()
        ret = mgr.dict()
        ctx = mp.get_context("spawn")
        ps = [ctx.Process(target=_worker_peer_dies, args=(r, 2, str(tmp_path / "uid.bin"), ret)) for r in range(2)]
        for p_ in ps:
            p_.start()
        for p_ in ps:
            p_.join(120)
            assert not p_.is_alive()
        out = dict(ret)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert out.get(1) == "left"
    st = out.get(0)
    assert isinstance(st, tuple) and st[0] == capi_status("CHIP_ERR_COMM") and st[1] < 30 and "fakerccl" in st[2], out


def capi_status(name):
    from cerebro_amd import capi
    return getattr(capi, name)
