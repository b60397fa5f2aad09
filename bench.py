#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X loop-detection core.

A "step" is one tick of Cerebro::descrip_N__dot__descrip_0_N (src/Cerebro.cpp:956-1100): the three newest
descriptors (rows l-1, l-2, l-3) against the DB prefix [0, l-50), top-k, accept rule -- on the workload
BASELINE.json's metric is quoted on: synthetic 4096-D fp32 descriptors x 1M keyframes.

  python bench.py --gpus N --steps K --warmup W
  N > 1 (torchrun, one rank per GPU): the 1M-row DB is row-sharded round-robin over the N GPUs (BASELINE
  config 4, strong scaling); each tick = local scan -> RCCL all-gather of 3 x top-k (score,index) per rank
  -> merge + decision on every rank.  The collective lives INSIDE libcerebro_hip.so (chip_comm_init_rank +
  ncclAllGather enqueued in-stream); torch.distributed is only the control plane (unique-id broadcast, barriers,
  max-over-ranks of the elapsed time).
  N > 1 WITHOUT torchrun (--gpus N, WORLD_SIZE unset): ONE process drives the N GPUs through chip_create_multi
  (the reference's process shape: one thread of one process, cerebro_node.cpp:499).

Prints ONE JSON line (rank 0).  `value` = ticks/s with the DB resident in HBM.  `roofline` prices the
dominant kernel (db_scan_topk) by its ALGORITHMIC bytes (4*D*rows scanned per launch) over its
mean duration measured with hipEvents on the kernel's own stream (chip_profile_*).  `cpu_baseline` times the
oracle's reference-faithful fp64 path (three separate GEMVs, single thread like the reference's
dot_product_th) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

D = 4096
LAG = 50
TOPK = 8
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
READ_CEILING_GBS = 7090.0  # best PURE-READ kernel in the scan's own access shape and occupancy (scripts/probes/hbm_read_probe.hip,
#                            profiles/r02_hbm_read_probe.txt; 7140 in any shape) -- context for roofline.frac, never its denominator
SEED = 20190412
UNIT_DATA = True     # rows of the synthetic DBs are unit-L2 vectors (chip_db_append_synthetic_unit); --data plain: round 1-5's generator


TICK_WINDOW = 1000   # distinct tick positions; longer runs cycle through them (keeps sharded ticks inside the replicated ring)


def plan_ticks(rows_scanned: int, n_ticks: int, avoid=()):
    """l_i = rows_scanned + LAG + 3*i ; every 4th tick is a planted revisit (fires the :1056 rule).
    avoid: (lo, hi) row windows the revisited rows p, p-1, p-2 must stay out of (rows planted by another plan)."""
    n_ticks = min(n_ticks, TICK_WINDOW)
    ls = [rows_scanned + LAG + 3 * i for i in range(n_ticks)]
    rng = np.random.default_rng(1)
    plants = []
    expect = []
    for i, l in enumerate(ls):
        if i % 4 == 0:
            p = int(rng.integers(1000, rows_scanned - 1000))
            while any(lo - 3 <= p < hi + 3 for lo, hi in avoid):
                p = int(rng.integers(1000, rows_scanned - 1000))
            for j in range(3):
                plants.append((l - 1 - j, p - j, 1))
            expect.append((l - 1, p))
        else:
            expect.append(None)
    return ls, sorted(plants), expect


def cpu_baseline(sample_cols: int, budget_s: float, kind: str = "eigen"):
    """Reference CPU path, one thread (the reference's dot_product_th is one thread; Eigen's GEMV is not parallel without OpenMP,
    which the reference does not enable): fp64 column-major M, 3 SEPARATE GEMVs + maxCoeff + last-index argmax
    (Cerebro.cpp:1023-1043).  kind "eigen": the Eigen-order port (oracle/dot_scan.c orc_ref_scan_f64_eigen_gemv3 -- Eigen 3.3's
    row-major GEMV for the reference's SSE2 Release build: four rows at a time, one Packet2d accumulator each, no FMA), the closest
    restatement of what the reference runs; kind "chain": the sequential-order port (one s += q[e]*col[e] chain per column).
    Returns columns/s, passes, seconds."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib  # checker/baseline only -- never on the product path
    M = oracle_lib.synth_rows(SEED, range(sample_cols + 3), D, unit=UNIT_DATA).astype(np.float64)
    v, vm, vmm = M[sample_cols + 2].copy(), M[sample_cols + 1].copy(), M[sample_cols].copy()
    scratch = (np.empty(sample_cols), np.empty(sample_cols), np.empty(sample_cols))
    if kind == "eigen":
        def scan(k):
            return oracle_lib.ref_scan_f64_eigen_gemv3(M, k, v, vm, vmm, 1, scratch)
    else:
        def scan(k):
            return oracle_lib.ref_scan_f64_colmajor(M, k, v, vm, vmm)
    scan(1000)  # touch
    n = 0
    t0 = time.perf_counter()
    while True:
        scan(sample_cols)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    cols_per_s = n * sample_cols / dt
    return cols_per_s, n, dt


def eigen_baseline(sample_cols: int, budget_s: float):
    """if THIS machine has Eigen (it is absent from the build image), time the literal Eigen statements of
    Cerebro.cpp:1026-1043 (oracle/eigen_probe.cc, built here with the reference's flags: g++ -O3 -DNDEBUG, no -march).
    Returns (cols_per_s, n, seconds, eigen_version) or None."""
    import glob
    import shutil
    import subprocess
    sys.path.insert(0, str(ROOT / "scripts"))
    try:
        from eigen_pin import find_eigen     # fixed include dirs, /opt, every site-packages / conda prefix (scripts/eigen_pin.py)
        hits = find_eigen()[0]
    except Exception:  # noqa: BLE001
        hits = []
    inc = hits[0] if hits else None
    gxx = shutil.which("g++")
    if inc is None or gxx is None:
        return None
    out = ROOT / "oracle" / "_build" / "eigen_probe"
    out.parent.mkdir(parents=True, exist_ok=True)
    try:
        r = subprocess.run([gxx, "-O3", "-DNDEBUG", "-std=c++11", "-I", inc, str(ROOT / "oracle" / "eigen_probe.cc"), "-o", str(out)],
                           capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            sys.stderr.write("[bench] Eigen found at " + inc + " but the probe did not build:\n" + r.stderr[-800:] + "\n")
            return None
        r = subprocess.run([str(out), str(D), str(sample_cols), str(budget_s)], capture_output=True, text=True, timeout=budget_s * 4 + 120)
        f = r.stdout.split()
        n, dt = int(f[f.index("ticks") + 1]), float(f[f.index("seconds") + 1])
        return n * sample_cols / dt, n, dt, f[1]
    except Exception as e:  # noqa: BLE001 -- a baseline must never take the benchmark down
        sys.stderr.write(f"[bench] Eigen probe failed: {e!r}\n")
        return None


def fmt_rows(n: int) -> str:
    return f"{n // 10**6}M" if n % 10**6 == 0 else f"{n // 1000}k" if n % 1000 == 0 else str(n)


def usable_cpus() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256
    logical CPUs behind a 16-CPU cpu.max; oversubscribing a CFS quota throttles instead of speeding up)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_all_cores(sample_cols: int, budget_s: float):
    """the same three Eigen-order GEMVs + maxCoeff + argmax with OpenMP over columns on every host core (what the
    reference would get from Eigen's OpenMP GEMV, which it does not enable).  Reported next to `cpu_baseline`, never as it."""
    import oracle_lib  # baseline only
    nthreads = usable_cpus()
    src = oracle_lib.synth_rows(SEED, range(2048), D, unit=UNIT_DATA).astype(np.float64)
    M = oracle_lib.tile_columns_omp(sample_cols, src, nthreads)           # first touch by the scanning threads
    v, vm, vmm = src[5].copy(), src[6].copy(), src[7].copy()
    scratch = (np.empty(sample_cols), np.empty(sample_cols), np.empty(sample_cols))
    oracle_lib.ref_scan_f64_eigen_gemv3(M, sample_cols, v, vm, vmm, nthreads, scratch)
    n = 0
    t0 = time.perf_counter()
    while True:
        oracle_lib.ref_scan_f64_eigen_gemv3(M, sample_cols, v, vm, vmm, nthreads, scratch)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    return n * sample_cols / dt, n, dt, nthreads


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: 256 CUs x 64 lanes x 2 flop x 2.4 GHz (v_fma_f64 at full rate)


def pnp_roofline(hyp_per_s: float, batch8_hyp_per_s: float):
    """SURVEY 8d prices the PnP kernels against the fp64 VECTOR peak.  Numerator: fp64 operations per hypothesis COUNTED from the
    solver (profiles/pnp_flops.json: an oracle build with -DORC_FLOP_COUNT over this leg's scene, scripts/count_pnp_flops.py) x
    hypotheses/s of the whole call (both kernels + host selection).  Context, from committed measurements: the peak this part
    sustains (profiles/r05_fp64_peak.txt: dense v_fma_f64 / the mul + add form a -ffp-contract=off build issues) and how busy the
    vector pipe is while the kernels run (profiles/pnp_pmc.json: SQ_INSTS_VALU x 4 cycles / (SIMDs x kernel cycles))."""
    fl = pm = pk = None
    try:
        fl = json.loads((ROOT / "profiles" / "pnp_flops.json").read_text())
    except Exception:
        pass
    try:
        pm = json.loads((ROOT / "profiles" / "pnp_pmc.json").read_text())
    except Exception:
        pass
    try:
        pk = json.loads([l for l in (ROOT / "profiles" / "r05_fp64_peak.txt").read_text().splitlines() if l.startswith("{")][-1])
    except Exception:
        pass
    fph = fl["flops_per_hypothesis"] if fl else None
    r = {"bound": "fp64-vector", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
         "achieved": hyp_per_s * fph / 1e12 if fph else None, "frac": hyp_per_s * fph / 1e12 / FP64_VECTOR_PEAK_TFLOPS if fph else None,
         "achieved_batch8": batch8_hyp_per_s * fph / 1e12 if fph else None,
         "frac_batch8": batch8_hyp_per_s * fph / 1e12 / FP64_VECTOR_PEAK_TFLOPS if fph else None,
         "flops_per_hypothesis": fph,
         "flops_per_hypothesis_dense_elimination": fl["flops_per_hypothesis_dense_elimination"] if fl else None,
         "flops_source": "profiles/pnp_flops.json: counted from the solver (oracle build -DORC_FLOP_COUNT over this scene), zero-multiplier rows of the "
                         "elimination skipped as the algorithm runs; _dense_elimination = without the skip (what the register-resident LU executes)",
         "traffic": None,
         "why_far_below_peak": "dependent fp64 chains (LU pivot column, Householder / Francis QR steps on a 27 x 27 matrix): one wave per hypothesis "
                               "in the eigen kernel, a lone wave issues one VALU instruction per ~8 cycles; <= 27 of 64 lanes active in the QR "
                               "phases; mul + add instead of fma (-ffp-contract=off, bit parity with the CPU path) halves the usable peak"}
    if pk:
        r["peak_measured_fma"] = pk.get("fp64_fma_tflops")
        r["peak_measured_mul_add"] = pk.get("fp64_mul_add_tflops")
        r["peak_measured_source"] = "profiles/r05_fp64_peak.txt (scripts/ubench/fp64_peak.hip: 8 independent chains per lane, 8 waves per SIMD)"
    if pm:
        r["valu_busy"] = pm.get("valu_busy")
        r["valu_insts_per_hypothesis"] = pm.get("valu_insts_per_hypothesis")
        r["valu_busy_source"] = pm.get("source")
        r["kernels"] = pm.get("kernels")
    return r


def icp_leg(chip):
    """Row N2: Umeyama-ICP-RANSAC (StaticTheiaPoseCompute::P3P_ICP's RANSAC branch, DlsPnpWithRansac.cpp:65-121) on 512 3-D/3-D
    correspondences: reference mode (<= 50 iterations), 1000 and 8000 fixed hypotheses, and the three-way pose of one loop candidate
    (PNP a->b + PNP b->a as one batched call with the ICP running underneath, Cerebro.cpp:1518,1572,1629)."""
    from cerebro_amd import capi
    from cerebro_amd.synth import make_scene, make_icp_scene
    A, B = make_icp_scene(N=512, outlier_frac=0.3, noise=0.01, seed=7)[:2]
    out = {"metric": "Umeyama-ICP-RANSAC hypotheses/sec (512 correspondences, 10-point samples, L2 error 0.1)", "unit": "hypotheses/s", "dtype": "f64"}
    for H, key in ((0, "reference_mode_ms_per_call"), (1000, "ms_per_call_1000_hyp"), (8000, "ms_per_call_8000_hyp")):
        p = capi.default_icp_params(); p.n_hypotheses = H; p.seed = 3
        for _ in range(3):
            chip.icp_ransac(A, B, p)
        n = 50
        t0 = time.perf_counter()
        for i in range(n):
            p.seed = 3 + i
            chip.icp_ransac(A, B, p)
        out[key] = 1e3 * (time.perf_counter() - t0) / n
    out["value"] = 8000 / (out["ms_per_call_8000_hyp"] / 1e3)
    # three-way pose of one loop candidate: the ICP is enqueued first and runs underneath the batched PnP pair
    scenes = [make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242 + i)[:2] for i in range(2)]
    pp = capi.default_ransac_params(); pp.n_hypotheses = 0
    pi = capi.default_icp_params(); pi.n_hypotheses = 0
    ts = []
    for i in range(53):
        t = time.perf_counter()
        chip.icp_ransac_enqueue(A, B, pi)
        chip.pnp_ransac_batch(scenes, pp, seeds=[300 + 2 * i, 301 + 2 * i])
        chip.icp_ransac_collect(A.shape[0])
        ts.append(time.perf_counter() - t)
    out["three_way_pose_ms"] = 1e3 * float(np.mean(ts[3:]))
    out["three_way_pose"] = "PNP(a->b) + PNP(b->a) (one batched call, <= 50 iterations each) with P3P_ICP's RANSAC underneath (enqueue / collect)"
    return out


def pnp_leg(chip, cpu_budget_s: float):
    """BASELINE config 3: 512 correspondences x 1000 hypotheses of 15 samples (DlsPnpWithRansac), whole call through
    chip_pnp_ransac (includes the 20 KB H2D of the correspondences and the host-side K7 selection)."""
    from cerebro_amd import capi
    from cerebro_amd.synth import make_scene
    X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    p = capi.default_ransac_params()
    p.n_hypotheses = 1000
    p.seed = 4242
    for _ in range(3):
        r = chip.pnp_ransac(X, uv, p)
    reps = 50
    t0 = time.perf_counter()
    for i in range(reps):
        p.seed = 4242 + i
        r = chip.pnp_ransac(X, uv, p)
    dt = time.perf_counter() - t0
    p.n_hypotheses = 0
    t1 = time.perf_counter()
    ref_times = []
    for i in range(reps):
        p.seed = 99 + i
        t_call = time.perf_counter()
        chip.pnp_ransac(X, uv, p)
        ref_times.append(time.perf_counter() - t_call)
    dt_ref = time.perf_counter() - t1
    # eight independent problems of the same shape per launch pair (chip_pnp_ransac_batch)
    scenes = [make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242 + i)[:2] for i in range(8)]
    p.n_hypotheses = 1000
    p.seed = 4242
    chip.pnp_ransac_batch(scenes, p)
    t3 = time.perf_counter()
    breps = 10
    for i in range(breps):
        chip.pnp_ransac_batch(scenes, p, seeds=[4242 + 8 * i + j for j in range(8)])
    dt_b = time.perf_counter() - t3
    # the production shape: a loop candidate is verified by TWO estimations with the roles swapped (Cerebro.cpp:1518, 1572), each in
    # reference mode -- here as one batched call (both problems in one launch pair)
    p.n_hypotheses = 0
    pair = [scenes[0], (scenes[1][0], scenes[1][1])]
    chip.pnp_ransac_batch(pair, p, seeds=[7, 8])
    pair_times = []
    for i in range(reps):
        t_call = time.perf_counter()
        chip.pnp_ransac_batch(pair, p, seeds=[200 + 2 * i, 201 + 2 * i])
        pair_times.append(time.perf_counter() - t_call)
    out = {"metric": "PnP-RANSAC hypotheses/sec (512 correspondences, 1000 hypotheses of 15 samples, DLS + L1 reprojection scoring)",
           "value": reps * 1000 / dt, "unit": "hypotheses/s", "ms_per_call_1000_hyp": 1e3 * dt / reps,
           "reference_mode_ms_per_call": 1e3 * dt_ref / reps, "reference_mode": "<=50 iterations, theia early termination",
           "reference_mode_ms_per_call_median": 1e3 * float(np.median(ref_times)), "reference_mode_ms_per_call_max": 1e3 * max(ref_times),
           "reference_mode_pair_ms_per_call": 1e3 * float(np.mean(pair_times)), "reference_mode_pair_ms_per_call_median": 1e3 * float(np.median(pair_times)),
           "reference_mode_pair": "the two role-swapped estimations of one loop candidate (Cerebro.cpp:1518,1572) as ONE batched call, <= 50 iterations each",
           "batch8_hypotheses_per_s": breps * 8 * 1000 / dt_b, "batch8_ms_per_call": 1e3 * dt_b / breps,
           "dtype": "f64", "n_models_last": r["summary"]["n_models"]}
    out["roofline"] = pnp_roofline(out["value"], out["batch8_hypotheses_per_s"])
    if cpu_budget_s > 0:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib   # cpu_baseline leg only
        n = 0
        t2 = time.perf_counter()
        while True:
            oracle_lib.pnp_ransac(X, uv, oracle_lib.ransac_params(n_hypotheses=1000, seed=4242 + n))
            n += 1
            d = time.perf_counter() - t2
            if d > cpu_budget_s or n >= 20:
                break
        out["cpu_baseline"] = {"value": n * 1000 / d, "unit": "hypotheses/s", "cores": 1, "kind": "port",
                               "sample": f"{n} oracle calls of 1000 hypotheses ({d:.1f} s), single thread"}
        nt = usable_cpus()
        prm = oracle_lib.ransac_params(n_hypotheses=1000, seed=4242)
        t4 = time.perf_counter()
        m = 0
        while True:
            best, _ = oracle_lib.pnp_hypotheses_mt(X, uv, prm, 4000, nt)
            m += 1
            d4 = time.perf_counter() - t4
            if d4 > cpu_budget_s or m >= 10:
                break
        out["cpu_baseline_all_cores"] = {"value": m * 4000 / d4, "unit": "hypotheses/s", "cores": nt, "kind": "port",
                                         "sample": f"{m} x 4000 oracle hypotheses, OpenMP over hypotheses ({d4:.1f} s)"}
    return out


def batch_leg(chip, rows: int, Q: int = 256):
    """Row N4: Q queries against the resident DB prefix in one pass -- fp32 GEMM on v_mfma_f32_32x32x2_f32 + fused top-k.
    MFMA-bound (arithmetic intensity Q/2 flop/B): priced against the 157.3 TFLOP/s dense fp32 matrix peak."""
    q = chip.read_rows((np.arange(Q, dtype=np.int64) * 7919) % rows)
    chip.query_batch(rows, q, TOPK)
    chip.profile_enable(True)
    chip.profile_reset()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        sc, ix = chip.query_batch(rows, q, TOPK)
    dt = (time.perf_counter() - t0) / n
    ms, cnt, _, _ = chip.profile_scan()
    chip.profile_enable(False)
    assert (ix[:, 0] == (np.arange(Q) * 7919) % rows).all()      # every query finds itself
    k_s = ms / 1e3 / cnt
    flops = 2.0 * Q * rows * D
    traffic = traffic_source = None
    pj = ROOT / "profiles" / "batch_traffic.json"
    if pj.exists() and D == 4096 and rows == 1_000_000 and Q == 256:
        try:
            traffic = json.loads(pj.read_text()).get("hbm_bytes_per_launch")
            traffic_source = "profiles/batch_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 of this launch shape, separate run)"
        except Exception:
            traffic = None
    return {"metric": "batched queries/sec (Q x DB fp32 GEMM on MFMA + fused top-k)", "value": Q / dt, "unit": "queries/s",
            "Q": Q, "db_rows": rows, "ms_per_call": dt * 1e3, "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact k-ordered fmaf chain)",
            "roofline": {"bound": "mfma", "achieved": flops / k_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": flops / k_s / 1e12 / 157.3, "kernel": "db_gemm_topk", "avg_kernel_ms": k_s * 1e3,
                         "algorithmic_flops_per_launch": flops, "traffic": traffic, "traffic_source": traffic_source}}


class c_stdout_to_stderr:
    """RCCL prints a version banner through C stdio when a communicator is created; with stdout on a pipe it would come out at
    process exit, AFTER the JSON line.  Creating communicators inside this context sends it to stderr instead (fd 1 is pointed at
    fd 2 and the C buffers are flushed before it is restored), so stdout carries the one JSON line and nothing else."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def call_with_deadline(fn, timeout_s):
    """Run fn() on a helper (daemon) thread and wait at most timeout_s for it.  Returns (finished, value, exception).  A helper
    that has not finished keeps whatever it is blocked on (an RCCL bootstrap, a collective that never completes): the caller moves
    on without it and must leave the process through os._exit."""
    box = {}

    def body():
        try:
            box["value"] = fn()
        except BaseException as e:   # noqa: BLE001 -- handed to the caller
            box["exc"] = e

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(timeout_s)
    return (not th.is_alive()), box.get("value"), box.get("exc")


def info_exchange(chip) -> int:
    return int(chip.info()["exchange"])


def run_ticks(chip, tick_ls, params, inflight, stamps=None):
    """Pipelined tick loop: up to `inflight` ticks enqueued ahead, results collected in order.  stamps (optional list) gets
    the host time of every collect -- the per-step cadence of the steady state."""
    from cerebro_amd import capi
    out = []
    W = max(1, min(inflight, capi.CHIP_MAX_INFLIGHT - 1))
    pending = []
    prev = -1
    for i, l in enumerate(tick_ls):
        if len(pending) == W:
            out.append(chip.loop_tick_collect(pending.pop(0)))
            if stamps is not None:
                stamps.append(time.perf_counter())
        s = i % W
        if l <= prev:
            chip.loop_reset()   # tick positions wrapped
        prev = l
        chip.loop_tick_enqueue(l, s, params)
        pending.append(s)
    while pending:
        out.append(chip.loop_tick_collect(pending.pop(0)))
        if stamps is not None:
            stamps.append(time.perf_counter())
    return out


def check_results(results, exp):
    """sanity (not the parity check -- that is tests/ -m gpu): planted revisits must fire with the planted index"""
    for r, e in zip(results, exp):
        if e is not None:
            assert r.found == 1 and r.idx_curr == e[0] and r.idx_prev == e[1], (r.as_dict(), e)
        else:
            assert r.found == 0, r.as_dict()


SIZES_TRAFFIC = None


def shape_name(rows: int, dim: int = 4096, storage: str = "f32") -> str:
    """10k / 29k / 100k for the BASELINE shape (4096-D float rows); 8192x29k, f64_1M for the reference's two production shapes."""
    return ("f64_" if storage == "f64" else "") + (f"{dim}x" if dim != 4096 else "") + fmt_rows(rows)


def sizes_traffic(rows: int, dim: int | None = None, storage: str = "f32"):
    """HBM-side bytes per launch of the size legs' launch shapes, from the committed PMC passes (profiles/scan_traffic_sizes.json:
    rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, scripts/gpu_scan_sizes_pmc.sh); None when that shape was not measured."""
    global SIZES_TRAFFIC
    if SIZES_TRAFFIC is None:
        try:
            SIZES_TRAFFIC = json.loads((ROOT / "profiles" / "scan_traffic_sizes.json").read_text())
        except Exception:
            SIZES_TRAFFIC = {}
    dim = D if dim is None else dim
    if dim == 4096 and storage == "f32":
        e = SIZES_TRAFFIC.get("sizes", {}).get(str(rows))
    else:
        e = SIZES_TRAFFIC.get("shapes", {}).get(shape_name(rows, dim, storage))
    return (e or {}).get("hbm_bytes_per_launch")


def size_leg(chip, rows, plan, params, inflight, n_ticks=240, warm=20, dim=None, elem_bytes=4):
    """The same tick loop over a SHORTER prefix of the resident DB (BASELINE configs 2 and 3: 10k and 100k keyframes).
    `plan` = (ls, expect) of plan_ticks(rows, ...).  Three figures per size, each named for what it is:
      * ms_per_step / frac_step : the pipelined loop (up to `inflight` ticks enqueued ahead; launches of consecutive ticks overlap
        on the tick streams) -- the sustained rate;
      * isolated_kernel_ms / frac_kernel (= roofline.achieved / .frac): ONE launch alone between two hipEvents on its stream --
        the per-kernel figure rocprofv3 reports for a synchronous tick (profiles/scan_traffic_sizes.json holds that trace);
      * sync_tick_us: host-to-host latency of one synchronous chip_loop_tick -- what a 10 Hz producer sees."""
    ls, expect = plan
    ls, expect = ls[:warm + n_ticks], expect[:warm + n_ticks]
    chip.loop_reset()
    run_ticks(chip, ls[:warm], params, inflight)
    chip.synchronize()
    t0 = time.perf_counter()
    res = run_ticks(chip, ls[warm:], params, inflight)
    chip.synchronize()
    dt = time.perf_counter() - t0
    check_results(res, expect[warm:])
    n = len(ls) - warm
    chip.loop_reset()
    chip.profile_enable(True)
    chip.profile_reset()
    run_ticks(chip, ls[warm:warm + 30], params, inflight)
    ms, cnt, _, _ = chip.profile_scan()
    chip.profile_enable(False)
    iso_s = ms / 1e3 / max(1, cnt)          # one launch ALONE on one stream, bracketed by hipEvents (profiled pass)
    step_s = dt / n                          # what a tick costs in the pipelined loop
    # synchronous ticks: enqueue + wait, one at a time (the live system's mode: dot_product_th ticks at 10 Hz, Cerebro.cpp:1100)
    chip.loop_reset()
    lat = []
    for i, l in enumerate(ls[warm:warm + 60]):
        t1 = time.perf_counter()
        r = chip.loop_tick(l, params)
        lat.append(time.perf_counter() - t1)
        assert r.status == 2
    lat = np.array(lat[10:])
    dim = D if dim is None else dim
    alg = 4.0 * dim * rows                                   # priced on the fp32 layout whatever the storage type
    actual = float(elem_bytes) * dim * rows                  # what the launch really has to read
    cache_resident = actual <= 256 * 2**20
    rows_form = actual <= 768 * 2**20
    return {"db_rows": rows, "value": n / dt, "unit": "loop-queries/s", "ms_per_step": 1e3 * step_s, "steps": n,
            "sync_tick_us": 1e6 * float(lat.mean()), "sync_tick_us_min": 1e6 * float(lat.min()),
            "roofline": {"bound": "hbm" if not cache_resident else "hbm (the 164 MB prefix is Infinity-Cache sized, 256 MiB: part of every re-read is served by the "
                                  "MALL, so `traffic` counts fabric requests, not DRAM bytes; a pure reader of such a buffer peaks at 7.3 TB/s on this part, profiles/r03_tick_timeline_10k.md)",
                         "achieved": alg / iso_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / iso_s / 1e9 / HBM_PEAK_GBS,
                         "frac_kernel": alg / iso_s / 1e9 / HBM_PEAK_GBS, "achieved_kernel": alg / iso_s / 1e9,
                         "frac_step": alg / step_s / 1e9 / HBM_PEAK_GBS, "achieved_step": alg / step_s / 1e9,
                         "priced_from": "achieved / frac / frac_kernel: algorithmic bytes / isolated_kernel_ms (one launch alone, hipEvents on its stream); "
                                        "achieved_step / frac_step: algorithmic bytes / ms_per_step of the pipelined loop, where the launches of consecutive "
                                        "ticks overlap on four tick streams (the ramp-up of one hides under the drain of the previous)",
                         "kernel_overlap_in_step": True,
                         "frac_kernel_actual_bytes": actual / iso_s / 1e9 / HBM_PEAK_GBS, "actual_bytes_per_launch": actual,
                         "traffic": sizes_traffic(rows, dim, "f64" if elem_bytes == 8 else "f32"),
                         "traffic_source": "profiles/scan_traffic_sizes.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this launch shape, synchronous ticks, separate run)"
                                           if sizes_traffic(rows, dim, "f64" if elem_bytes == 8 else "f32") is not None else None,
                         "kernel": "db_scan_topk_rows (row-batched form, R = 1, temporal loads: prefixes <= 768 MiB; the synchronous and the pipelined "
                                   "tick run it fused: one launch per tick)" if rows_form else
                                   ("db_scan_topk_rows (R = 2, non-temporal loads: rows of >= 32 KiB up to 2 GiB, fused)" if dim * elem_bytes >= 32768 and actual <= 2048 * 2**20 else
                                    "db_scan_topk (pipelined ticks and this isolated launch)" + ("; synchronous ticks: db_scan_topk_rows fused, R = 1, non-temporal loads (<= 4 GiB)" if actual <= 4096 * 2**20 else "")),
                         "isolated_kernel_ms": iso_s * 1e3, "launches": cnt, "algorithmic_bytes_per_launch": alg, "cache_resident": cache_resident}}


def shape_leg(rows, dim, storage, inflight, n_ticks=240):
    """The reference's two PRODUCTION shapes on a ctx of their own: the default model emits 8192-D float32
    descriptors into a DB capped at 29 000 columns (Cerebro.cpp:946,1021; whole_image_desc_compute_server.py:206-218), the only 4096-D
    model (ReljaNetVLAD, server.py:148-149) emits genuine float64 => double rows.  Same tick loop, same three figures as size_leg; the
    roofline is priced on 4*D*k (never inflated by the storage type) AND on the bytes the launch really reads."""
    from cerebro_amd import capi
    ls, plants, expect = plan_ticks(rows, 20 + n_ticks)
    params = capi.default_dot_params()
    with capi.Chip(dim, capacity_hint=ls[-1], storage=(None if storage == "f32" else "f64")) as c:
        c.append_synthetic(ls[-1], SEED, plants, unit=UNIT_DATA)
        c.synchronize()
        leg = size_leg(c, rows, (ls, expect), params, inflight, n_ticks=n_ticks, dim=dim, elem_bytes=8 if storage == "f64" else 4)
    leg["D"] = dim
    leg["storage"] = storage
    return leg


def paced_tick_leg(rows, n=60, pause_ms=100, spin=False):
    """Synchronous ticks at the reference's own cadence -- dot_product_th ticks at 10 Hz (Cerebro.cpp:1100), so the GPU has idled for
    100 ms when a tick arrives -- launched and through the resident instance, measured by the C caller examples/sync_tick_latency.cc
    (built by `make` into cerebro_amd/lib/): through the ctypes binding the first call after a sleep carries tens of microseconds of
    cold-interpreter jitter on about every third tick (scripts/gpu_paced_ticks.py), which is not the library's.  None without the tool.
    spin: the caller BUSY-WAITS through the pause instead of sleeping -- the GPU idles exactly as long, the host core stays awake: what
    is left of the idle penalty then is the GPU's (the cold hardware queue), not the waking host's (profiles/r06_paced.md)."""
    import subprocess
    tool = ROOT / "cerebro_amd" / "lib" / "sync_tick_latency"
    if not tool.exists():
        return None
    out = {"rows": rows, "ticks": n, "pause_ms": pause_ms, "pause": "spin" if spin else "sleep", "caller": "examples/sync_tick_latency.cc"}
    for mode in ("launched", "resident"):
        env = {k: v for k, v in os.environ.items() if k != "CHIP_TICK_RESIDENT"}
        if mode == "resident":
            env["CHIP_TICK_RESIDENT"] = "1"
        try:
            r = subprocess.run([str(tool), str(rows), str(n), "0", str(pause_ms), "spin" if spin else "sleep"], env=env, capture_output=True, text=True, timeout=120)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
            out[mode] = json.loads(line)["sync_tick"]
        except Exception as e:      # noqa: BLE001 -- a reported figure, not a gate
            out[mode] = {"error": str(e)[:200]}
    return out


def resident_leg(rows, n_ticks=400, warm=40):
    """Synchronous ticks through the resident scan instance (opt-in mode CHIP_TICK_RESIDENT=1: the tick is a 64-byte command to a
    kernel that stays on the chip, no launch) next to the same ticks launched, on a ctx of its own with `rows` + 400 synthetic rows:
    host-to-host latency of chip_loop_tick through this (ctypes) binding, records compared byte for byte."""
    from cerebro_amd import capi
    p = capi.default_dot_params()
    p.min_new = -(1 << 30)
    ls = [rows + 50 + 3 * (i % 100) for i in range(warm + n_ticks)]
    out = {}
    for mode in ("launched", "resident"):
        old = os.environ.pop("CHIP_TICK_RESIDENT", None)
        if mode == "resident":
            os.environ["CHIP_TICK_RESIDENT"] = "1"
        try:
            with capi.Chip(D, capacity_hint=rows + 500) as c:
                c.append_synthetic(rows + 400, 777, [], unit=UNIT_DATA)
                recs, lat = [], []
                for l in ls:
                    t1 = time.perf_counter()
                    r = c.loop_tick(l, p)
                    lat.append(time.perf_counter() - t1)
                    recs.append(bytes(r))
                lat = np.array(lat[warm:])
                out[mode] = {"recs": recs, "mean_us": 1e6 * float(lat.mean()), "p50_us": 1e6 * float(np.median(lat)), "min_us": 1e6 * float(lat.min())}
        finally:
            os.environ.pop("CHIP_TICK_RESIDENT", None)
            if old is not None:
                os.environ["CHIP_TICK_RESIDENT"] = old
    same = out["launched"].pop("recs") == out["resident"].pop("recs")
    if not same:
        raise SystemExit(f"bench: resident-instance ticks differ from launched ticks at {rows} rows")
    return {"rows": rows, "launched": out["launched"], "resident": out["resident"], "records_identical": same, "ticks": n_ticks}

CONFIG_KEY_CAP = 24   # the driver's record keeps the first 24 keys of `config` (BENCH_r05.json: everything appended later was dropped)

# (key in the record, path into the legs) -- in PRIORITY order: the PnP half of BASELINE's metric first, then the isolated-launch
# roofline fractions of every prefix size / production shape, then the tick latencies at the reference's cadence.
HEADLINE_SCALARS = [
    ("pnp_hyp_per_s", "pnp.value"),
    ("pnp_batch8_hypotheses_per_s", "pnp.batch8_hypotheses_per_s"),
    ("pnp_reference_mode_ms_per_call", "pnp.reference_mode_ms_per_call"),
    ("pnp_reference_mode_pair_ms_per_call", "pnp.reference_mode_pair_ms_per_call"),
    ("pnp_roofline_frac", "pnp.roofline.frac"),
    ("pnp_roofline_frac_batch8", "pnp.roofline.frac_batch8"),
    ("size_10k_roofline_frac_kernel", "sizes.10k.roofline.frac_kernel"),
    ("size_29k_roofline_frac_kernel", "sizes.29k.roofline.frac_kernel"),
    ("size_100k_roofline_frac_kernel", "sizes.100k.roofline.frac_kernel"),
    ("size_8192x29k_roofline_frac_kernel", "shapes.8192x29k.roofline.frac_kernel"),
    ("size_f64_1M_roofline_frac_kernel", "shapes.f64_1M.roofline.frac_kernel"),
    ("size_f64_1M_roofline_frac_actual_bytes", "shapes.f64_1M.roofline.frac_kernel_actual_bytes"),
    ("size_10k_sync_tick_10hz_launched_us", "paced_10hz.first.launched.p50_us"),
    ("size_10k_sync_tick_10hz_resident_us", "paced_10hz.first.resident.p50_us"),
    ("size_10k_sync_tick_us", "sizes.10k.sync_tick_us"),
    ("size_8192x29k_sync_tick_us", "shapes.8192x29k.sync_tick_us"),
]
# scalars that ride in `roofline` only (after the ones above): everything else the round-5 record carried as config.* keys
ROOFLINE_EXTRA_SCALARS = [
    ("size_29k_sync_tick_us", "sizes.29k.sync_tick_us"),
    ("size_f64_1M_ms_per_step", "shapes.f64_1M.ms_per_step"),
    ("size_10k_sync_tick_10hz_launched_after_contexts_us", "paced_10hz.after_contexts.launched.p50_us"),
    ("size_10k_sync_tick_10hz_resident_after_contexts_us", "paced_10hz.after_contexts.resident.p50_us"),
    ("size_10k_sync_tick_10hz_spin_launched_us", "paced_10hz.spin.launched.p50_us"),
    ("size_10k_sync_tick_10hz_spin_resident_us", "paced_10hz.spin.resident.p50_us"),
    ("size_10k_sync_tick_launched_us", "resident_tick.10k.launched.p50_us"),
    ("size_10k_sync_tick_resident_us", "resident_tick.10k.resident.p50_us"),
    ("size_29k_sync_tick_launched_us", "resident_tick.29k.launched.p50_us"),
    ("size_29k_sync_tick_resident_us", "resident_tick.29k.resident.p50_us"),
    ("pnp_ms_per_call_1000_hyp", "pnp.ms_per_call_1000_hyp"),
    ("pnp_flops_per_hypothesis", "pnp.roofline.flops_per_hypothesis"),
    ("pnp_valu_insts_per_hypothesis_build_solve", "pnp.roofline.valu_insts_per_hypothesis.pnp_build_solve"),
    ("pnp_valu_busy_build_solve", "pnp.roofline.valu_busy.pnp_build_solve"),
    ("pnp_valu_busy_eig_score", "pnp.roofline.valu_busy.pnp_eig_score"),
    ("pnp_cpu_baseline_hyp_per_s", "pnp.cpu_baseline.value"),
    ("icp_hyp_per_s", "icp.value"),
    ("icp_three_way_pose_ms", "icp.three_way_pose_ms"),
    ("batch256_queries_per_s", "batch.value"),
    ("batch256_roofline_frac", "batch.roofline.frac"),
]


def _dig(src, path):
    v = src
    for part in path.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None


def finalize_record(out: dict) -> dict:
    """Lay the one JSON line out so that the DRIVER's record keeps both halves of BASELINE's metric (VERDICT r5 missing 1).  The driver
    keeps the top-level contract keys, the scalar entries of `roofline`, and the first CONFIG_KEY_CAP keys of `config`; whole legs
    (`pnp`, `sizes`, `shapes`, `icp`, `batch`, `resident_tick`, `paced_10hz`) are listed as extra keys and dropped.  So the legs'
    headline scalars are repeated (a) inside `roofline`, right after the contract's own keys and before any string, and (b) as the
    FIRST keys of `config` after `workload`; `config` is cut to CONFIG_KEY_CAP keys, what does not fit moves to `config_more`.
    Pure function of `out` (tests/test_bench_plan.py pins the ordering and the count on a synthetic record)."""
    head = {k: _dig(out, path) for k, path in HEADLINE_SCALARS}
    head = {k: v for k, v in head.items() if v is not None}
    extra = {k: _dig(out, path) for k, path in ROOFLINE_EXTRA_SCALARS}
    extra = {k: v for k, v in extra.items() if v is not None}

    roof = out["roofline"]
    contract = ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    own_scalars = [k for k, v in roof.items() if k not in contract and isinstance(v, (int, float)) and not isinstance(v, bool)]
    own_text = [k for k, v in roof.items() if k not in contract and k not in own_scalars and not isinstance(v, dict)]
    new_roof = {k: roof.get(k) for k in contract}
    for k in own_scalars:
        new_roof[k] = roof[k]
    new_roof.update(head)
    new_roof.update(extra)
    for k in own_text:
        new_roof[k] = roof[k]
    nested = {k: v for k, v in roof.items() if isinstance(v, dict)}
    if nested:
        out["roofline_context"] = nested
    out["roofline"] = new_roof

    cfg = out["config"]
    new_cfg = {"workload": cfg["workload"]}
    new_cfg.update(head)
    for k, v in cfg.items():
        new_cfg.setdefault(k, v)
    if len(new_cfg) > CONFIG_KEY_CAP:
        keys = list(new_cfg)
        out["config_more"] = {k: new_cfg[k] for k in keys[CONFIG_KEY_CAP:]}
        new_cfg = {k: new_cfg[k] for k in keys[:CONFIG_KEY_CAP]}
    out["config"] = new_cfg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=1_000_000, help="DB rows scanned per tick (k)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="columns in the CPU baseline sample")
    ap.add_argument("--dim", type=int, default=4096, help="descriptor size (BASELINE: 4096; the reference's default model emits 8192)")
    ap.add_argument("--inflight", type=int, default=16)
    ap.add_argument("--storage", choices=["f32", "f64"], default="f32",
                    help="row type of the DB: f32 (BASELINE: synthetic fp32 descriptors) or f64 (double rows, e.g. ReljaNetVLAD)")
    ap.add_argument("--data", choices=["unit", "plain"], default="unit",
                    help="synthetic rows: unit = exactly unit-L2 (SURVEY 8d, the default); plain = integers x one constant (rounds 1-5; norm spread 1.1 %% rms)")
    ap.add_argument("--no-pnp", action="store_true", help="skip the auxiliary PnP-RANSAC leg (config 3)")
    ap.add_argument("--no-batch", action="store_true", help="skip the auxiliary many-query MFMA leg (row N4)")
    ap.add_argument("--no-sizes", action="store_true", help="skip the 10k / 100k legs (BASELINE configs 2, 3)")
    ap.add_argument("--replicated", action="store_true",
                    help="N > 1 only: every GPU holds the whole DB and serves its own stream of ticks (no collective; weak "
                         "scaling) instead of the default N-way row shard of BASELINE config 4")
    ap.add_argument("--force-sharded", action="store_true",
                    help="testing aid: run the sharded code path (scan -> local merge -> RCCL all-gather -> merge) even with 1 rank")
    ap.add_argument("--force-group", action="store_true",
                    help="testing aid for a 1-GPU box: run the one-process group ctx (chip_create_multi, ncclCommInitAll) even with --gpus 1")
    ap.add_argument("--same-device", action="store_true",
                    help="testing aid for a 1-GPU box: one process, --gpus N sub-contexts all on device 0 (device-copy exchange)")
    ap.add_argument("--host-exchange", action="store_true",
                    help="N > 1 under torchrun: exchange through torch.distributed (cerebro_amd/sharded.py) instead of the in-library RCCL")
    ap.add_argument("--no-shapes", action="store_true",
                    help="skip the legs over the reference's production shapes (8192-D x 29k float rows, 4096-D x 1M double rows; ctxs of their own, 34 GB)")
    ap.add_argument("--paced-ticks", type=int, default=40,
                    help="N = 1: synchronous ticks at the reference's 10 Hz cadence per mode and position (0 = skip; each costs 0.1 s)")
    args = ap.parse_args()
    global UNIT_DATA
    UNIT_DATA = args.data == "unit"

    # The paced (10 Hz) tick leg runs FIRST, in a process of its own, before this process has touched the GPU: round 5's record showed
    # 69.6 us for a launched tick at 10 Hz where the standalone tool on an otherwise idle box reads 52 us, and blamed the bench
    # process's held contexts.  Running it here, and again once every context exists (after_contexts), measures that blame.
    paced_first = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_sizes and not args.force_sharded and not args.force_group \
            and args.storage == "f32" and args.paced_ticks > 0 and args.rows >= 1_000_000:
        paced_first = paced_tick_leg(10_000, n=args.paced_ticks)

    import gc
    import torch
    from cerebro_amd import capi
    # A generation-2 collection of a process that has imported torch walks ~10^6 objects: 35-50 ms.  Round 4 caught one landing inside
    # the 50-call reference-mode PnP loop (ONE call of 38 ms among 0.52 ms calls: the leg read 1.2-1.6 ms per call whenever the size legs
    # had run before it and shifted the allocation count).  Timed regions must not contain the collector: freeze what exists, switch it off.
    gc.collect()
    gc.freeze()
    gc.disable()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    group_mode = world == 1 and (args.gpus > 1 or args.force_group)   # one process drives all GPUs (chip_create_multi)
    if world > 1 and world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    dist = None
    replicated = args.replicated and world > 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # Control plane only (unique id, barriers, max of the elapsed time).  "gloo" keeps torch from building a second
        # RCCL communicator next to the library's; BENCH_DIST_BACKEND=nccl selects torch's RCCL backend instead.
        # With --host-exchange the data path goes through this backend too (then nccl is the default).
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl" if args.host_exchange else "gloo")
        ndev = torch.cuda.device_count()
        if ndev < world:
            local_rank = local_rank % ndev   # functional-test aid: ranks share devices (only valid with the host exchange over gloo)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    n_ticks = args.warmup + args.steps
    # the 10k / 100k legs (N = 1) tick over shorter prefixes of the same DB: their planted query rows are DB rows of the
    # longer scans, so the longer plans keep their revisited rows out of those windows
    # (29k rows = the reference's own capacity, Cerebro.cpp:946)
    leg_rows = [r for r in (10_000, 29_000, 100_000) if r + 4000 < args.rows] if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 \
        and not args.no_sizes and not args.force_sharded and not args.force_group else []
    windows = [(r, r + LAG + 3 * 260 + 3) for r in leg_rows]
    ls, plants, expect = plan_ticks(args.rows, n_ticks, avoid=windows)
    total_rows = ls[-1]
    if n_ticks > len(ls):   # cycle: tick i uses position i % TICK_WINDOW (the host-side last_l is reset at every wrap)
        ls = [ls[i % TICK_WINDOW] for i in range(n_ticks)]
        expect = [expect[i % TICK_WINDOW] for i in range(n_ticks)]
    size_plans = {}
    for r in leg_rows:
        l2, p2, e2 = plan_ticks(r, 260, avoid=[w for w in windows if w[0] < r])
        plants = sorted(plants + p2)
        size_plans[r] = (l2, e2)
    assert len({p[0] for p in plants}) == len(plants)

    global D
    D = args.dim
    storage = None if args.storage == "f32" else "f64"
    abandon_process = False
    group_fallback = None
    if group_mode:
        ndev = torch.cuda.device_count()
        devices = [0] * args.gpus if args.same_device else list(range(args.gpus))
        if not args.same_device and ndev < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {ndev} device(s) visible (use --same-device for a functional run)")

        def make_group(copy_exchange):
            if os.environ.get("BENCH_HANG_GROUP_CREATE") and not copy_exchange:   # test hook: a create that never returns
                time.sleep(1e6)
            return capi.Chip(D, capacity_hint=total_rows, devices=devices, storage=storage, copy_exchange=copy_exchange)

        # chip_create_multi -> ncclCommInitAll is a blocking bootstrap.  The library runs it under its own deadline
        # (CHIP_COMM_INIT_TIMEOUT_MS) and falls back to the device-copy exchange; the create as a whole runs under a second, outer
        # deadline here, so that whatever hangs inside it costs this run its RCCL exchange, not its JSON line.
        os.environ.setdefault("CHIP_COMM_INIT_TIMEOUT_MS", str(int(1000 * float(os.environ.get("BENCH_COMM_INIT_TIMEOUT", "180")))))
        with c_stdout_to_stderr():
            fin, chip, exc = call_with_deadline(lambda: make_group(False), float(os.environ.get("BENCH_GROUP_CREATE_TIMEOUT", float(os.environ.get("BENCH_COMM_INIT_TIMEOUT", "180")) + 60.0)))
        if not fin or exc is not None:
            why = "did not return within its deadline" if not fin else f"failed: {exc}"
            sys.stderr.write(f"[bench] chip_create_multi over RCCL {why}: rebuilding the group with the device-copy exchange\n")
            group_fallback = "create hung" if not fin else "create failed"
            abandon_process = not fin
            chip = make_group(True)
        elif chip.info().get("comm_init_abandoned"):
            abandon_process = True      # a helper thread of the library is still inside ncclCommInitAll
            group_fallback = "ncclCommInitAll hung (abandoned at the library's deadline)"
    else:
        chip = capi.Chip(D, capacity_hint=total_rows, device=local_rank, shard_rank=0 if replicated else rank,
                         shard_count=1 if replicated else world, storage=storage)
    det = None
    det_over_torch_nccl = False
    exchange = "none"
    if (world > 1 and not replicated) or args.force_sharded:
        if args.host_exchange:
            if dist is None:
                import torch.distributed as dist
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            from cerebro_amd.sharded import ShardedLoopDetector
            det = ShardedLoopDetector(chip, topk=TOPK, device=torch.device("cuda", local_rank))
            exchange = f"host-driven: torch.distributed all_gather_into_tensor ({dist.get_backend()})"
        else:
            # in-library RCCL: rank 0 makes the unique id, the control plane hands it round, every rank attaches
            uid = [capi.comm_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            # The attach is a blocking RCCL bootstrap: it runs on a helper thread with a deadline, so that a rank stuck in it
            # (a node whose RCCL bootstrap network is unusable) costs the benchmark its exchange, not the whole run.
            attach = {}

            def attach_comm():
                try:
                    if os.environ.get("BENCH_FAIL_COMM_INIT"):   # test hooks: exercise the agreed fallbacks below
                        raise capi.ChipError(capi.CHIP_ERR_COMM, "chip_comm_init_rank", "BENCH_FAIL_COMM_INIT")
                    if os.environ.get("BENCH_HANG_COMM_INIT"):
                        time.sleep(1e6)
                    chip.comm_init_rank(uid[0], world, rank)
                    attach["ok"] = True
                except capi.ChipError as e:   # e.g. RCCL refusing the topology
                    attach["err"] = e
                    # the library's own deadline (CHIP_COMM_INIT_TIMEOUT_MS) fired: ncclCommInitRank is still running on its helper
                    attach["hung"] = bool(chip.info().get("comm_init_abandoned"))

            os.environ.setdefault("CHIP_COMM_INIT_TIMEOUT_MS", str(int(1000 * float(os.environ.get("BENCH_COMM_INIT_TIMEOUT", "180")))))
            with c_stdout_to_stderr():
                th = threading.Thread(target=attach_comm, daemon=True)
                th.start()
                th.join(float(os.environ.get("BENCH_COMM_INIT_TIMEOUT", "180")) + 30.0)
            comm_hung = th.is_alive() or bool(attach.get("hung"))
            ok = 1 if attach.get("ok") else 0
            if comm_hung:
                sys.stderr.write(f"[bench rank {rank}] chip_comm_init_rank did not return within its deadline\n")
            elif not ok:
                sys.stderr.write(f"[bench rank {rank}] chip_comm_init_rank failed: {attach.get('err')}\n")
            if dist is not None:   # every rank must learn of it and take the same path
                flag = torch.tensor([ok, 0 if comm_hung else 1], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok, comm_hung = int(flag[0].item()), int(flag[1].item()) == 0
            if ok:
                exchange = "in-library RCCL: ncclAllGather of 3 x top-k (score, index) per rank per tick, enqueued in-stream"
            elif comm_hung:
                # an RCCL bootstrap that hangs would hang torch.distributed's nccl group as well: exchange over the control
                # plane that demonstrably works.  The stuck ctx is abandoned (its attach is still running), the process leaves
                # through os._exit after the JSON line.
                sys.stderr.write(f"[bench rank {rank}] falling back to the host-driven exchange over {dist.get_backend() if dist else 'nccl'}\n")
                abandon_process = True
                chip = capi.Chip(D, capacity_hint=total_rows, device=local_rank, shard_rank=rank, shard_count=world, storage=storage)
                from cerebro_amd.sharded import ShardedLoopDetector
                if dist is None:
                    import torch.distributed as dist
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
                    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
                    dist.init_process_group("gloo")
                det = ShardedLoopDetector(chip, topk=TOPK, device=torch.device("cuda", local_rank))
                exchange = f"host-driven fallback: torch.distributed all_gather_into_tensor ({dist.get_backend()}) after chip_comm_init_rank hung"
            else:
                # agreed fallback: the host-driven exchange over a torch.distributed RCCL group (the ctx of a rank whose attach
                # did succeed is rebuilt, so that every rank runs the same code path)
                sys.stderr.write(f"[bench rank {rank}] falling back to the host-driven exchange (torch.distributed nccl)\n")
                chip.close()
                chip = capi.Chip(D, capacity_hint=total_rows, device=local_rank, shard_rank=rank, shard_count=world, storage=storage)
                from cerebro_amd.sharded import ShardedLoopDetector
                if dist is None:
                    import torch.distributed as dist
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
                    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                grp = dist.new_group(backend="nccl") if dist.get_backend() != "nccl" else None
                det = ShardedLoopDetector(chip, topk=TOPK, group=grp, device=torch.device("cuda", local_rank))
                det_over_torch_nccl = dist.get_backend() != "nccl"   # a second RCCL bootstrap (torch's): its first collective runs under the warmup deadline too
                exchange = "host-driven fallback: torch.distributed all_gather_into_tensor (nccl) after chip_comm_init_rank failed"
    elif group_mode:
        exchange = {capi.CHIP_EXCHANGE_RCCL: "in-library RCCL (ncclCommInitAll, one worker thread per device)",
                    capi.CHIP_EXCHANGE_COPY: "in-library device copies" + (" (devices repeat: RCCL refuses two ranks on one device)" if args.same_device else
                                                                          f" -- FALLBACK, RCCL was asked for: {group_fallback or 'ncclCommInitAll failed: ' + str(chip.last_comm_error())}")}[chip.info()["exchange"]]
    params = capi.default_dot_params()

    def fill(c):
        t = time.perf_counter()
        c.append_synthetic(total_rows, SEED, plants, unit=UNIT_DATA)
        return time.perf_counter() - t

    t_fill = fill(chip)

    def barrier():
        # device-wide synchronisation only while nothing of this process is stuck on the device: after a fallback an abandoned
        # collective may still sit in a stream of the old ctx, and torch.cuda.synchronize() would wait for it for ever
        if not abandon_process:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        chip.synchronize()

    def make_run(chip, det):
        if det is None:
            def run(tick_ls):
                return run_ticks(chip, tick_ls, params, args.inflight)
        else:
            def run(tick_ls):
                # host-driven exchange: scan(i+1) on the ctx's scan streams overlaps all-gather(i) + merge(i) on torch's stream
                out = []
                W = max(1, min(args.inflight, capi.CHIP_MAX_INFLIGHT - 1))
                pending = []
                prev = -1
                for i, l in enumerate(tick_ls):
                    if len(pending) == W:
                        out.append(det.collect(pending.pop(0)))
                    s = i % W
                    if l <= prev:
                        chip.loop_reset()   # tick positions wrapped
                    prev = l
                    st = det.tick_enqueue(l, s, params)
                    assert st == capi.CHIP_TICK_SCANNED
                    pending.append(s)
                while pending:
                    out.append(det.collect(pending.pop(0)))
                return out
        return run

    run = make_run(chip, det)
    chip.loop_reset()
    barrier()
    # Warmup.  On a multi-GPU exchange inside the library it is also the FIRST execution of the collective (ncclAllGather over
    # xGMI between real devices): it runs under a deadline, and if it does not come back on some rank every rank rebuilds its
    # context on an exchange that needs no RCCL -- the run still ends with a JSON line, and the line says what happened.
    guarded = (det is None and ((world > 1 and not replicated) or args.force_sharded or group_mode) and info_exchange(chip) != capi.CHIP_EXCHANGE_NONE) \
        or det_over_torch_nccl

    def warm():
        if os.environ.get("BENCH_HANG_WARMUP") and guarded and (info_exchange(chip) == capi.CHIP_EXCHANGE_RCCL or det_over_torch_nccl):   # test hook
            time.sleep(1e6)
        return run(ls[:args.warmup])

    runtime_fallback = None
    if guarded:
        fin, _, exc = call_with_deadline(warm, float(os.environ.get("BENCH_WARMUP_TIMEOUT", "120")))
        okw = 1 if (fin and exc is None) else 0
        if dist is not None:
            flag = torch.tensor([okw], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            okw = int(flag[0].item())
        if not okw:
            why = "did not finish within its deadline" if not fin else (f"failed: {exc}" if exc is not None else "failed on another rank")
            sys.stderr.write(f"[bench rank {rank}] warmup over the {'torch.distributed nccl' if det_over_torch_nccl else 'in-library'} exchange {why}: rebuilding on a fallback exchange\n")
            abandon_process = True            # the old ctx (and whatever is stuck in it) is left alone
            if group_mode:
                chip = capi.Chip(D, capacity_hint=total_rows, devices=devices, storage=storage, copy_exchange=True)
                runtime_fallback = f"in-library device copies -- FALLBACK: the warmup over RCCL {why}"
            else:
                chip = capi.Chip(D, capacity_hint=total_rows, device=local_rank, shard_rank=rank, shard_count=world, storage=storage)
                from cerebro_amd.sharded import ShardedLoopDetector
                if dist is None:
                    import torch.distributed as dist
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
                    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
                    dist.init_process_group("gloo")
                det = ShardedLoopDetector(chip, topk=TOPK, device=torch.device("cuda", local_rank))   # default group: the control plane (gloo)
                runtime_fallback = f"host-driven fallback: torch.distributed all_gather_into_tensor ({dist.get_backend()}) -- the warmup over " \
                                   f"{'torch.distributed nccl (itself the fallback of a failed chip_comm_init_rank)' if det_over_torch_nccl else 'in-library RCCL'} {why}"
                det_over_torch_nccl = False
            exchange = runtime_fallback
            t_fill = fill(chip)
            run = make_run(chip, det)
            chip.loop_reset()
            run(ls[:args.warmup])
    else:
        warm()
    info = chip.info()
    wanted_rccl = ((world > 1 and not replicated) or args.force_sharded or (group_mode and not args.same_device)) and not args.host_exchange
    rccl_ranks = int(info.get("comm_ranks", 0)) if runtime_fallback is None else 0   # ncclCommCount of the communicator the library's exchange runs over
    exchange_fallback = bool(wanted_rccl and rccl_ranks != (args.gpus if group_mode else world))
    barrier()
    t0 = time.perf_counter()
    results = run(ls[args.warmup:])
    barrier()
    elapsed = time.perf_counter() - t0

    # Roofline pass (outside the timed region): the same launches again with hipEvents bracketing every db_scan_topk
    # launch on the stream it runs on.  Kept out of the throughput loop because each timing event is a barrier packet
    # between back-to-back scans (~5 us each).
    n_prof = min(args.steps, 30)
    chip.loop_reset()
    chip.profile_enable(True)
    chip.profile_reset()
    run(ls[args.warmup:args.warmup + n_prof])
    barrier()
    scan_ms, n_launch, bytes_last, span_ms = chip.profile_scan()
    chip.profile_enable(False)

    check_results(results, expect[args.warmup:])

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_gpus = args.gpus if group_mode else world
        shards = 1 if replicated else n_gpus
        local_rows = (args.rows + shards - 1) // shards
        alg_bytes = 4.0 * D * local_rows                       # one pass of the prefix as fp32 (this GPU's share), whatever the storage type
        avg_s = scan_ms / 1e3 / max(1, n_launch)              # per-launch duration (what rocprofv3 --stats reports)
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = traffic_source = None
        pj = ROOT / "profiles" / "scan_traffic.json"
        # the committed PMC measurement is of the headline launch (4096-D x 1M fp32 rows on one GPU); other shapes report null
        if pj.exists() and D == 4096 and args.rows == 1_000_000 and n_gpus == 1 and args.storage == "f32":
            try:
                traffic = json.loads(pj.read_text()).get("hbm_bytes_per_launch")
                traffic_source = "profiles/scan_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this launch shape, separate run; not re-measured in this run)"
            except Exception:
                traffic = None
        out = {
            "metric": f"loop-queries/sec (ticks of 3 descriptors vs {D}-D x {fmt_rows(args.rows)} DB)",   # same string at every N; PnP-RANSAC hypotheses/sec: see "pnp" (N = 1)
            "value": (world if replicated else 1) * args.steps / elapsed,   # replicas each run `steps` ticks of their own
            "unit": "loop-queries/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if replicated else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("synthetic, SURVEY 8d: unit-L2 rows (on-device generator, seed 20190412: Irwin-Hall integers, sum of squares exact in "
                     "integers, element = (float)(v / sqrt(S)) -- CPU oracle and GPU generate identical bits; | |row| - 1 | < 2e-7), planted "
                     "revisits = unit(5 src + noise), cos 0.98" if UNIT_DATA else
                     "synthetic (on-device integer-domain generator, seed 20190412, planted revisits; rows are unit-norm in expectation, "
                     "norm spread ~1.1 % rms at D=4096: Irwin-Hall integers x one constant, so that CPU and GPU generate identical bits)"),
            # <= CONFIG_KEY_CAP keys once finalize_record has put the legs' headline scalars in front (N = 1: 8 keys here; N > 1: no legs)
            "config": ({"workload": f"{D}-D fp32 descriptors x {args.rows} keyframe DB, 3 queries/tick, top-{TOPK} + accept rule",
                        "db_rows": args.rows, "D": D, "queries_per_tick": 3, "topk": TOPK,
                        "storage": "fp32 rows, fp64 accumulate" if args.storage == "f32" else "fp64 rows (2x the HBM bytes; roofline priced on 4*D*k)",
                        "sharding": "single GPU", "exchange": exchange}
                       if (n_gpus == 1 and world == 1 and not args.force_sharded and not args.force_group) else
                       {"workload": f"{D}-D fp32 descriptors x {args.rows} keyframe DB, 3 queries/tick, top-{TOPK} + accept rule",
                        "db_rows": args.rows, "D": D, "queries_per_tick": 3, "topk": TOPK,
                        "storage": "fp32 rows, fp64 accumulate" if args.storage == "f32" else "fp64 rows (2x the HBM bytes; roofline priced on 4*D*k)",
                        "process_layout": "one process" if world == 1 else f"{world} processes (one per GPU)",
                        "exchange": exchange,
                        "rccl_ranks": rccl_ranks,                    # 0 = no RCCL communicator inside the library
                        "exchange_fallback": exchange_fallback,      # True: RCCL was asked for and did NOT carry the exchange of all ranks
                        "exchange_runtime_fallback": runtime_fallback,   # not None: the first collective (warmup) hung / failed and the run was rebuilt on another exchange
                        "comm_init_abandoned": bool(info.get("comm_init_abandoned")) or abandon_process,   # something of this process is still stuck in RCCL: it leaves through os._exit
                        "control_plane": (dist.get_backend() if dist is not None else None),
                        "sharding": "single GPU" if n_gpus == 1 else (f"{world} replicas of the whole DB, independent tick streams, no collective" if replicated
                                                                      else f"row round-robin over {n_gpus} GPUs + all-gather of top-k"),
                        "descriptor_queries_per_s": 3 * args.steps / elapsed,
                        "preflight": os.environ.get("BENCH_PREFLIGHT"),   # scripts/preflight_8gpu.sh: verdict of the device-count-gated tests, run before this line
                        "db_fill_s": t_fill, "arch": info["arch"], "n_cus": info["n_cus"]}),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "db_scan_topk", "avg_kernel_ms": avg_s * 1e3, "launches": n_launch,
                         "measured": "hipEvents around each launch on the kernel's stream, separate pass of the same ticks",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "pure_read_ceiling": {"value": READ_CEILING_GBS, "unit": "GB/s", "frac_of_it": achieved / READ_CEILING_GBS,
                                               "source": "profiles/r02_hbm_read_probe.txt: do-nothing reader in the scan's access shape, another box of the pool"}},
        }
        if size_plans:
            out["sizes"] = {fmt_rows(r): size_leg(chip, r, plan, params, args.inflight) for r, plan in sorted(size_plans.items())}
            out["sizes"][fmt_rows(args.rows)] = {"db_rows": args.rows, "value": out["value"], "unit": "loop-queries/s", "ms_per_step": out["ms_per_step"],
                                                 "steps": args.steps,
                                                 "roofline": out["roofline"]}
        single = n_gpus == 1 and world == 1 and not args.force_sharded and not args.force_group
        out["details"] = {"storage": "fp32 rows (verified-lossless narrowing of the f64 wire type), fp64 accumulate" if args.storage == "f32"
                                     else "fp64 rows (double-row mode: 2x the HBM bytes of the fp32 layout; roofline still priced on 4*D*k algorithmic bytes)",
                          "loop_query": "one tick of Cerebro::descrip_N__dot__descrip_0_N = 3 descriptor queries + top-k + accept rule",
                          "descriptor_queries_per_s": 3 * args.steps / elapsed, "db_fill_s": t_fill, "arch": info["arch"], "n_cus": info["n_cus"],
                          "rccl_ranks": rccl_ranks, "exchange_fallback": exchange_fallback, "exchange_runtime_fallback": runtime_fallback,
                          "test_hooks": info.get("test_hooks")}
        if paced_first is not None:
            # the paced leg ran BEFORE this process created any HIP context (first) and runs again now, with the
            # bench's contexts, streams and 16+ GB of device memory alive (after_contexts) -- the pair says what the held queues cost
            out["paced_10hz"] = {"first": paced_first, "after_contexts": paced_tick_leg(10_000, n=args.paced_ticks),
                                 "spin": paced_tick_leg(10_000, n=max(10, args.paced_ticks // 2), spin=True),
                                 "what": "synchronous chip_loop_tick at the reference's 10 Hz cadence (100 ms idle before every tick), C caller "
                                         "examples/sync_tick_latency.cc in a process of its own; `first` = nothing else on the GPU, `after_contexts` = "
                                         "while this bench process holds its contexts (DB, tick streams, PnP / ICP / batch buffers); `spin` = the caller "
                                         "busy-waits through the 100 ms instead of sleeping (same GPU idle time, host core awake): the difference to the "
                                         "slept figures is the waking host, what remains above the back-to-back tick is the idle hardware queue"}
        if size_plans and single and args.storage == "f32" and (capi.load_library().chip_build_scan_forms() & 2):
            # the opt-in resident scan instance at the two sizes of the reference's operating range (ctxs of their own, a few hundred ms)
            out["resident_tick"] = {fmt_rows(r): resident_leg(r) for r in (10_000, 29_000)}
        if size_plans and single and args.storage == "f32" and D == 4096 and args.rows >= 1_000_000 and not args.no_shapes:
            # the reference's two production shapes, each on a ctx of its own
            out["shapes"] = {shape_name(29_000, 8192): shape_leg(29_000, 8192, "f32", args.inflight),
                             shape_name(1_000_000, 4096, "f64"): shape_leg(1_000_000, 4096, "f64", args.inflight, n_ticks=60)}
        if single and not args.no_pnp:
            out["pnp"] = pnp_leg(chip, min(args.cpu_budget, 5.0))
            out["icp"] = icp_leg(chip)
        if single and not args.no_batch and args.storage == "f32":
            out["batch"] = batch_leg(chip, args.rows)
        if n_gpus == 1 and world == 1 and args.cpu_budget > 0:
            eig = eigen_baseline(args.cpu_sample, min(args.cpu_budget, 10.0))
            cols_per_s, n, dt = cpu_baseline(args.cpu_sample, args.cpu_budget * 2 / 3, "eigen")
            if eig is not None:   # the reference's own Eigen statements, on this host
                out["cpu_baseline_eigen"] = {"value": eig[0] / args.rows, "unit": "loop-queries/s", "cores": 1, "kind": "eigen",
                                             "sample": f"{eig[1]} ticks of the literal Eigen {eig[3]} statements (Cerebro.cpp:1026-1043) over a "
                                                       f"{args.cpu_sample}-column x {D} MatrixXd ({eig[2]:.1f} s), g++ -O3 -DNDEBUG without -march "
                                                       f"(the reference's Release flags), scaled to {args.rows} columns"}
            out["eigen_probe"] = "Eigen found: cpu_baseline_eigen is the reference's own statements" if eig is not None else \
                "no <Eigen/Dense> on this host (searched EIGEN3_INCLUDE_DIR, the system include dirs, /opt/*, every Python site-packages / conda prefix: scripts/eigen_pin.py, profiles/r05_eigen_pin.json): cpu_baseline is the Eigen-order port"
            host = f"host has {os.cpu_count()} logical CPUs ({usable_cpus()} usable under the cgroup quota)"
            out["cpu_baseline"] = {"value": cols_per_s / args.rows, "unit": "loop-queries/s", "cores": 1, "kind": "port",
                                   "port_of": "eigen-order port: Eigen 3.3 row-major GEMV as the reference's SSE2 Release build runs it (four rows at a "
                                              "time, one Packet2d accumulator each, mul + add, predux, scalar tail; oracle/dot_scan.c "
                                              "orc_ref_scan_f64_eigen_gemv3, bit-identical to the order emulation: tests/test_oracle_eigen_order.py)",
                                   "sample": f"{n} ticks of 3 separate fp64 GEMVs + maxCoeff + last-index argmax over a {args.cpu_sample}-column x "
                                             f"{D} column-major M ({dt:.1f} s), scaled to {args.rows} columns; single thread like the reference's "
                                             f"dot_product_th (Eigen without OpenMP); {host}"}
            cols_c, n_c, dt_c = cpu_baseline(args.cpu_sample, args.cpu_budget / 3, "chain")
            out["cpu_baseline_chain"] = {"value": cols_c / args.rows, "unit": "loop-queries/s", "cores": 1, "kind": "port",
                                         "sample": f"{n_c} ticks ({dt_c:.1f} s) of the sequential-order port (one s += q[e]*col[e] chain per column, -O2 "
                                                   "-ffp-contract=off: does not vectorise) -- the secondary figure, the slowest faithful restatement"}
            ac_cols = max(args.cpu_sample, 200_000)      # 6.5 GB of fp64: large enough to defeat the host caches
            cols_per_s, n, dt, nt = cpu_baseline_all_cores(ac_cols, min(args.cpu_budget, 6.0))
            out["cpu_baseline_all_cores"] = {"value": cols_per_s / args.rows, "unit": "loop-queries/s", "cores": nt, "kind": "port",
                                             "sample": f"{n} ticks over a {ac_cols}-column x {D} fp64 M ({dt:.1f} s), the same Eigen-order port with OpenMP "
                                                       f"static over columns, scaled to {args.rows} columns; {os.cpu_count()} logical CPUs visible, "
                                                       f"{nt} usable under the affinity mask / cgroup quota"}
        print(json.dumps(finalize_record(out)), flush=True)

    # whatever C stdio output is still buffered in this process (RCCL banners of torch's own communicator, ...) leaves through
    # stderr: stdout has carried the JSON line and must carry nothing after it
    if abandon_process:   # a helper thread is still inside RCCL's bootstrap: no orderly teardown is possible
        if dist is not None:
            dist.barrier()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    with c_stdout_to_stderr():
        if det is not None:
            det.close()
        chip.close()
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
