"""GPU tests of the resident scan instance (opt-in, CHIP_TICK_RESIDENT=1: cerebro_amd/csrc/kernels.hip db_scan_resident,
chip_api.hip resident_*): synchronous ticks over cache-sized prefixes are COMMANDS to a kernel that stays on the chip instead of
launches.  The bar is the one of every other tick path -- the 64-byte decision record byte for byte -- against the two-launch path
(CHIP_TICK_FUSED=0: ordinary kernel-boundary visibility) and, for one tick, against the CPU oracle; plus the life cycle: an instance
whose lease ran out is replaced, appended rows are seen, a new DB segment retires it, other kernels of the ctx run next to it, ticks
beyond its prefix bound and pipelined ticks take the launched path, and destroying the ctx sends it home."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import oracle_lib
from cerebro_amd import capi, synth

pytestmark = pytest.mark.gpu
D = 4096


def resident_stats(chip):
    fn = chip.lib.chip_debug_resident_stats
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    t, n = C.c_int64(0), C.c_int64(0)
    assert fn(chip.h, C.byref(t), C.byref(n)) == 0
    return t.value, n.value


def every_tick_params():
    p = capi.default_dot_params()
    p.min_new = -(1 << 30)          # every tick runs, whatever the previous l was
    return p


def test_resident_ticks_equal_launched_ticks_and_oracle(monkeypatch):
    seed, n_rows = 424242, 12_100
    ls = [3_003, 7_777, 10_050, 10_053, 10_056, 11_000, 12_001, 12_100]
    plants = [(10_049 - j, 5_000 - j, 1) for j in range(3)] + [(12_000 - j, 2_000 - j, 1) for j in range(3)] + [(5_005, 5_000, 2)]
    p = every_tick_params()
    monkeypatch.setenv("CHIP_TICK_FUSED", "0")
    monkeypatch.delenv("CHIP_TICK_RESIDENT", raising=False)
    with capi.Chip(D, capacity_hint=n_rows) as ref:
        ref.append_synthetic(n_rows, seed, plants)
        want = {l: bytes(ref.loop_tick(l, p)) for l in ls}
        assert resident_stats(ref) == (0, 0)
    assert sum(capi.TickResult.from_buffer_copy(w).found for w in want.values()) >= 2
    monkeypatch.delenv("CHIP_TICK_FUSED")
    monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
    monkeypatch.setenv("CHIP_RESIDENT_LEASE_MS", "40")
    n_ticks = int(os.environ.get("CHIP_RESIDENT_TICKS", "60000"))
    with capi.Chip(D, capacity_hint=n_rows) as chip:
        chip.append_synthetic(n_rows, seed, plants)
        bad = 0
        for i in range(n_ticks):
            l = ls[(i * 5) % len(ls)]
            bad += bytes(chip.loop_tick(l, p)) != want[l]
        assert bad == 0
        ticks, launches = resident_stats(chip)
        assert ticks == n_ticks and launches >= 1
        # one record against the CPU oracle as well (prefix l - 50, queries l-1..l-3)
        l = 10_050
        r = chip.loop_tick(l, p)
        qrows = oracle_lib.synth_rows(seed, [l - 1, l - 2, l - 3], D, plants)
        wsc, wix = oracle_lib.scan_topk_synth(seed, l - 50, D, qrows, 1, plants, nthreads=os.cpu_count() or 1)
        assert list(r.argmax) == list(wix[:, 0]) and [float(x).hex() for x in r.maxv] == [float(x).hex() for x in wsc[:, 0]]
        assert r.found == 1 and r.idx_curr == l - 1 and r.idx_prev == 5_005   # the later exact duplicate wins the tie (Cerebro.cpp:1035-1043)
        # the lease runs out (40 ms without a command): the instance leaves by itself, the next tick launches another
        before = resident_stats(chip)[1]
        time.sleep(0.4)
        for l in ls:
            assert bytes(chip.loop_tick(l, p)) == want[l]
        assert resident_stats(chip)[1] >= before + 1      # (more only if the box stalled the loop for another lease)
        # pipelined ticks: one command at a time -- the second enqueue takes the launched path; both records are right
        chip.loop_tick_enqueue(ls[2], 0, p)
        chip.loop_tick_enqueue(ls[3], 1, p)
        assert bytes(chip.loop_tick_collect(1)) == want[ls[3]] and bytes(chip.loop_tick_collect(0)) == want[ls[2]]
        t1 = resident_stats(chip)[0]
        assert t1 == n_ticks + 1 + len(ls) + 1


def test_resident_sees_appends_new_segments_and_other_kernels(monkeypatch):
    """Rows appended while the instance is alive are scanned and used as queries by the next tick; a tick beyond the resident bound
    (512 MiB prefix) is launched; opening a new DB segment (capacity exceeded) retires the instance and the following tick gets a new
    one; a PnP-RANSAC call of the same ctx runs next to the waiting instance."""
    seed = 99
    n0, n1, n2 = 9_000, 10_500, 40_000
    rng = np.random.default_rng(5)
    extra = rng.standard_normal((n1 - n0, D)).astype(np.float32)
    extra /= np.linalg.norm(extra, axis=1, keepdims=True)
    extra[-2] = extra[100]                       # a revisit among the appended rows: query row n1-2 == appended row n0+100
    extra[-3] = extra[99]
    extra[-4] = extra[98]
    p = every_tick_params()

    def run(resident):
        if resident:
            monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
            monkeypatch.setenv("CHIP_RESIDENT_LEASE_MS", "3000")
        else:
            monkeypatch.delenv("CHIP_TICK_RESIDENT", raising=False)
        out = []
        with capi.Chip(D, capacity_hint=n1) as chip:
            chip.append_synthetic(n0, seed, [])
            out.append(bytes(chip.loop_tick(n0, p)))
            chip.append_f32(extra)
            out.append(bytes(chip.loop_tick(n1 - 1, p)))           # queries n1-2, n1-3, n1-4: the planted revisit
            out.append(bytes(chip.loop_tick(n1, p)))
            stats_a = resident_stats(chip)
            X, uv, _, _ = synth.make_scene(N=256, seed=3)
            chip.pnp_ransac(X, uv, capi.default_ransac_params())   # (its first call allocates: the instance is retired for that)
            chip.loop_tick(n1, p)
            chip.pnp_ransac(X, uv, capi.default_ransac_params())   # ... the second runs next to the waiting instance
            out.append(bytes(chip.loop_tick(n1, p)))
            chip.append_synthetic(n2 - n1, seed + 1, [])           # far beyond capacity_hint: new segments
            stats_b = resident_stats(chip)
            out.append(bytes(chip.loop_tick(10_000, p)))           # resident again (new instance)
            out.append(bytes(chip.loop_tick(n2, p)))               # 655 MB prefix: launched
            out.append(bytes(chip.loop_tick(9_500, p)))
            stats_c = resident_stats(chip)
        return out, stats_a, stats_b, stats_c

    want, *_ = run(False)
    got, sa, sb, sc = run(True)
    assert got == want
    r = capi.TickResult.from_buffer_copy(got[1])
    assert r.found == 1 and r.idx_prev == n0 + 100
    assert sa[0] == 3 and sa[1] >= 1
    assert sc[0] == sa[0] + 4 and sc[1] >= sa[1] + 2   # the ticks after the PnP allocation and after the new segment each found no instance


def test_resident_at_the_reference_capacity(monkeypatch):
    """29k rows (the reference's own capacity, Cerebro.cpp:946): the instance runs the prefix with one workgroup per CU and rows
    claimed within the workgroup (the launched tick: two workgroups per CU) -- same records, and the ticks went through it."""
    seed, n_rows = 31, 29_400
    ls = [20_000, 25_003, 29_000, 29_003, 29_399, 29_400]
    plants = [(28_999 - j, 14_000 - j, 1) for j in range(3)] + [(29_398 - j, 50 - j, 1) for j in range(3)]
    p = every_tick_params()
    monkeypatch.delenv("CHIP_TICK_RESIDENT", raising=False)
    with capi.Chip(D, capacity_hint=n_rows) as ref:
        ref.append_synthetic(n_rows, seed, plants)
        want = {l: bytes(ref.loop_tick(l, p)) for l in ls}
    assert sum(capi.TickResult.from_buffer_copy(w).found for w in want.values()) == 2
    monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
    with capi.Chip(D, capacity_hint=n_rows) as chip:
        chip.append_synthetic(n_rows, seed, plants)
        n = int(os.environ.get("CHIP_RESIDENT_TICKS_29K", "3000"))
        bad = 0
        for i in range(n):
            l = ls[(i * 5) % len(ls)]
            bad += bytes(chip.loop_tick(l, p)) != want[l]
        assert bad == 0
        assert resident_stats(chip)[0] == n


def test_resident_ticks_while_another_thread_appends_across_segments(monkeypatch):
    """The live system's two threads: the dot-product thread ticks (resident instance) while the descriptor thread appends -- here past
    a segment boundary (32 768 rows of 4096 floats), which retires the instance while ticks are in flight and makes the ticking thread
    launch new ones.  Every tick over the old rows must return the record it returned before the appends."""
    import threading
    seed, n0 = 4711, 12_000
    ls = [5_000, 9_999, 10_050, 11_990]
    plants = [(10_049 - j, 3_000 - j, 1) for j in range(3)]
    p = every_tick_params()
    monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
    monkeypatch.setenv("CHIP_RESIDENT_LEASE_MS", "50")
    with capi.Chip(D, capacity_hint=n0) as chip:
        chip.append_synthetic(n0, seed, plants)
        want = {l: bytes(chip.loop_tick(l, p)) for l in ls}
        assert capi.TickResult.from_buffer_copy(want[10_050]).found == 1
        stop = threading.Event()
        errors = []

        def appender():
            try:
                for i in range(24):
                    chip.append_synthetic(2_000, seed + 1 + i, [])      # 12k -> 60k rows: crosses into a second segment
                    time.sleep(0.01)
            except Exception as e:      # noqa: BLE001
                errors.append(e)
            finally:
                stop.set()

        th = threading.Thread(target=appender)
        th.start()
        n = bad = 0
        while not stop.is_set() or n < 2000:
            l = ls[n % len(ls)]
            bad += bytes(chip.loop_tick(l, p)) != want[l]
            n += 1
        th.join()
        assert not errors, errors
        assert bad == 0, (bad, n)
        ticks, launches = resident_stats(chip)
        # (round 6: ticks that arrive while the appender grows the DB by a segment -- the mode is PAUSED for that section -- are launched)
        assert n + len(ls) - 200 <= ticks <= n + len(ls) and launches >= 2, (ticks, n, launches)
        # ... and a tick whose queries and prefix lie in the NEW segment (launched: 0.9 GB), then a resident one again
        r = chip.loop_tick(60_000, p)
        assert r.status == capi.CHIP_TICK_SCANNED
        assert bytes(chip.loop_tick(10_050, p)) == want[10_050]


def test_resident_command_that_workgroup_0_never_saw_is_recovered(monkeypatch, hooks_lib):
    """A command posted in the moment the lease runs out can reach the other workgroups while workgroup 0 is already leaving: they run
    it, the ticket never fills, the instance goes.  The test hook reproduces exactly that (the 5th command is not written into
    workgroup 0's line); the collecting call must notice the exit, send the stragglers home, post the command again to a new instance
    and return the right record."""
    seed, n_rows = 5, 11_000
    ls = [10_050, 10_053, 9_000, 10_990]
    plants = [(10_049 - j, 1_000 - j, 1) for j in range(3)]
    p = every_tick_params()
    monkeypatch.delenv("CHIP_TICK_RESIDENT", raising=False)
    with capi.Chip(D, capacity_hint=n_rows) as ref:
        ref.append_synthetic(n_rows, seed, plants)
        want = {l: bytes(ref.loop_tick(l, p)) for l in ls}
    monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
    monkeypatch.setenv("CHIP_RESIDENT_LEASE_MS", "30")
    monkeypatch.setenv("CHIP_TEST_RESIDENT_SKIP_MASTER", "5")
    with capi.Chip(D, capacity_hint=n_rows) as chip:
        chip.append_synthetic(n_rows, seed, plants)
        took = []
        for i in range(12):
            l = ls[i % len(ls)]
            t0 = time.perf_counter()
            assert bytes(chip.loop_tick(l, p)) == want[l], i
            took.append(time.perf_counter() - t0)
        ticks, launches = resident_stats(chip)
        assert ticks == 12
        fn = chip.lib.chip_debug_resident_mode
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
        mode = fn(chip.h)
        if mode == 2:                          # direct lines (large PCIe BAR): the 5th tick waited for the lease and a new instance
            assert launches >= 2 and took[4] > 0.02 and float(np.median(took[5:])) < 0.02, (launches, took)
        else:                                  # one command line for all: nothing to skip, nothing to recover
            assert launches >= 1


def test_pause_resume_and_frees_under_a_10hz_style_tick_stream(monkeypatch):
    """resident_stop in front of a hipFree did not keep the tick thread from relaunching the instance; with ticks
    arriving faster than the lease the device-wide frees of pnp_reserve / icp_reserve then waited for as long as ticks kept coming.
    Now those sections PAUSE the mode (no instance is launched until they are over; ticks in between are ordinary launches), and the
    same pause is public (chip_resident_pause / chip_resident_resume) for device-wide calls of other libraries in the process."""
    import threading
    seed, n_rows = 5, 11_000
    ls = [10_050, 10_053, 9_000, 10_990]
    p = every_tick_params()
    monkeypatch.delenv("CHIP_TICK_RESIDENT", raising=False)
    with capi.Chip(D, capacity_hint=n_rows) as ref:
        ref.append_synthetic(n_rows, seed, [])
        want = {l: bytes(ref.loop_tick(l, p)) for l in ls}
    monkeypatch.setenv("CHIP_TICK_RESIDENT", "1")
    monkeypatch.setenv("CHIP_RESIDENT_LEASE_MS", "2000")          # far longer than any gap between the ticks below: it never leaves by itself
    with capi.Chip(D, capacity_hint=n_rows) as chip:
        chip.append_synthetic(n_rows, seed, [])
        for l in ls:
            assert bytes(chip.loop_tick(l, p)) == want[l]
        t0, n0 = resident_stats(chip)
        assert t0 == len(ls) and n0 == 1
        # the public pause: ticks are launched, records identical, no new instance; nested pauses need as many resumes
        chip.resident_pause(); chip.resident_pause()
        for l in ls:
            assert bytes(chip.loop_tick(l, p)) == want[l]
        assert resident_stats(chip) == (t0, n0)
        chip.resident_resume()
        assert bytes(chip.loop_tick(ls[0], p)) == want[ls[0]] and resident_stats(chip) == (t0, n0)
        chip.resident_resume()
        assert bytes(chip.loop_tick(ls[0], p)) == want[ls[0]]
        assert resident_stats(chip) == (t0 + 1, n0 + 1)
        # a tick thread hammering the ctx (faster than any lease) while the geometry thread's buffers grow three times (pnp_reserve and
        # icp_reserve free and reallocate ~ten device / pinned buffers each): bounded, and every record still right
        stop, bad, n_ticks = threading.Event(), [], [0]

        def ticker():
            i = 0
            while not stop.is_set():
                l = ls[i % len(ls)]
                if bytes(chip.loop_tick(l, p)) != want[l]:
                    bad.append(l)
                i += 1
            n_ticks[0] = i

        th = threading.Thread(target=ticker)
        th.start()
        try:
            t_start = time.perf_counter()
            for N in (64, 700, 2600):
                X, uv = synth.make_scene(N=N, outlier_frac=0.2, noise_px=0.5, seed=N)[:2]
                rp = capi.default_ransac_params(); rp.n_hypotheses = 64; rp.seed = 3
                assert chip.pnp_ransac(X, uv, rp)["summary"]["best_hypothesis"] >= 0
                A, B = synth.make_icp_scene(N=N, outlier_frac=0.2, noise=0.01, seed=N)[:2]
                ip = capi.default_icp_params(); ip.n_hypotheses = 64; ip.seed = 3
                chip.icp_ransac(A, B, ip)
            took = time.perf_counter() - t_start
        finally:
            stop.set()
            th.join(60)
        assert not th.is_alive() and not bad and n_ticks[0] > 50
        assert took < 20.0, took                                   # (before: unbounded -- each free waited for an instance the ticks kept alive)
        t1, n1 = resident_stats(chip)
        assert t1 > t0 + 1                                         # the mode came back after every paused section
