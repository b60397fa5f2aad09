"""EuRoC-shaped surrogate runs for BASELINE configs 1 and 5 (the real thing -- EuRoC MH-01..05 bags, the NetVLAD weights and a recorded
`loopcandidates_liverun.json` of the reference -- is absent from this image).

What the surrogate keeps of a real run, because the candidate selection depends on it:
  * descriptors are a unit-norm AR(1) walk  d_t = normalize(a d_{t-1} + sqrt(1 - a^2) n_t)  over the camera frames: temporal
    neighbours are CORRELATED (a = 0.9 fast motion ... 0.9999 the platform standing still), so the best score of a tick and the
    scores of its temporal neighbours are 1e-2 ... 1e-5 apart -- the argmax is close, unlike in random-vector fixtures;
  * 5-10 revisits per sequence: a stretch of frames re-traverses an earlier stretch (forward, backward or at another speed) with
    similarity b in {0.97 ... 0.845}: strong loops, borderline ones around DOT_PROD_THRESH = 0.85f, and triples whose three argmax
    straddle LOCALITY_THRESH = 12 (src/Cerebro.cpp:1056);
  * not every frame is a keyframe, and keyframes are dropped by the reference's dynamic skip rule (src/Cerebro.cpp:189-203:
    skip_frac = 1 - incoming_diff_ms / estimated_descriptor_compute_time_ms, `rand()`), so DB rows are irregular in time;
  * the dot-product thread ticks at ~10 Hz with occasional stalls against descriptors arriving at <= 20 Hz: `l` grows by 0 ... 7
    between ticks (most ticks see < 3 new rows and do nothing, src/Cerebro.cpp:962-966);
  * time stamps are EuRoC's (MH_01 starts at 1403636579.763555584, 20 Hz); `global_a / global_b` of the recorded dump are indices
    into the map of ALL frames (src/Cerebro.cpp:1143-1144), not DB rows.
All floating-point work that decides bits is done by the C oracle (oracle/surrogate.c orc_ar1_step, oracle/dot_scan.c synthetic
noise rows), all random choices by an integer LCG here, so the same run is regenerated bit for bit on any machine; the committed
fixtures (tests/golden/euroc_surrogate_*.json, written by tests/golden/make_euroc_surrogate.py) pin its descriptors by SHA-256 and
hold the "recorded reference run": the candidate list the reference's arithmetic produces (Eigen 3.3 SSE2 GEMV order,
orc_loop_tick_order(order = 1)) in the reference's dump format (src/Cerebro.cpp:1127-1164, src/cerebro_node.cpp:769-770).
Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import hashlib

import numpy as np

import oracle_lib

D = 4096
FRAME_NS = 50_000_000                                  # 20 Hz camera
T0 = (1403636579, 763555584)                           # first image of EuRoC MH_01_easy
SEQ_FRAMES = {"mh01": [3682], "mh01_05": [3682, 3040, 2700, 2033, 2273]}     # images per EuRoC machine-hall sequence
SEQ_GAP_S = 30                                          # pause between the bags of the merged replay


class Lcg:
    """ANSI C example generator (the one cerebro_replay uses for the clique policy): integers only"""

    def __init__(self, seed=1):
        self.s = seed & 0xFFFFFFFF

    def rand(self) -> int:                               # 0 .. 32767
        self.s = (self.s * 1103515245 + 12345) & 0xFFFFFFFF
        return (self.s >> 16) & 0x7FFF

    def below(self, n: int) -> int:
        return self.rand() % n

    def pick(self, seq):
        return seq[self.below(len(seq))]


def _ar1(prev, noise_row, alpha, round_f32, out):
    lib = oracle_lib.load()
    lib.orc_ar1_step.restype = None
    lib.orc_ar1_step(None if prev is None else prev.ctypes.data_as(C.c_void_p), noise_row.ctypes.data_as(C.c_void_p),
                     C.c_double(alpha), C.c_int32(out.size), C.c_int32(1 if round_f32 else 0), out.ctypes.data_as(C.c_void_p))


def plan(variant: str, seed: int):
    """Integer-only plan of a run: per frame (alpha, revisit source or -1, beta), keyframe flags, DB rows, tick schedule."""
    rng = Lcg(seed)
    frames_per_seq = SEQ_FRAMES[variant]
    F = sum(frames_per_seq)
    alpha = np.empty(F)
    src = np.full(F, -1, dtype=np.int64)
    beta = np.zeros(F)
    stamps_ns = np.empty(F, dtype=np.int64)
    t_ns = T0[0] * 10**9 + T0[1]
    f0 = 0
    revisits = []
    for si, nf in enumerate(frames_per_seq):
        for i in range(nf):
            stamps_ns[f0 + i] = t_ns + i * FRAME_NS
        t_ns += nf * FRAME_NS + SEQ_GAP_S * 10**9
        # motion segments: 40-200 frames each with one alpha
        i = 0
        while i < nf:
            n = 40 + rng.below(161)
            a = rng.pick([0.9, 0.95, 0.97, 0.98, 0.99, 0.99, 0.995, 0.999, 0.9999])
            alpha[f0 + i:f0 + min(nf, i + n)] = a
            i += n
        # revisits: later sequences of the merged run re-traverse earlier ones more often (the same machine hall)
        n_rev = 7 + rng.below(4) + (3 if si > 0 else 0)
        for _ in range(3 * n_rev):
            if sum(1 for r0, _ in revisits if r0 >= f0) >= n_rev:
                break
            L = 50 + rng.below(120)
            start = f0 + 300 + rng.below(max(1, nf - 300 - L)) if si == 0 else f0 + 20 + rng.below(max(1, nf - 20 - L))
            lo_src = 5
            hi_src = start - 200                           # well behind the 50-row lag
            if hi_src - L - lo_src < 10:
                continue
            s0 = lo_src + rng.below(hi_src - L - lo_src)
            b = rng.pick([0.97, 0.95, 0.93, 0.90, 0.88, 0.86, 0.852, 0.848, 0.845])
            mode = rng.pick(["fwd", "fwd", "back", "fast", "slow"])
            if any(not (start + L <= r0 or r0 + rl <= start) for r0, rl in revisits):
                continue
            revisits.append((start, L))
            for j in range(L):
                if start + j >= f0 + nf:
                    break
                sj = {"fwd": s0 + j, "back": s0 + L - 1 - j, "fast": s0 + min(2 * j, 2 * L - 1) // 1, "slow": s0 + j // 2}[mode]
                sj = min(sj, start - 150)
                src[start + j] = sj
                beta[start + j] = b
        f0 += nf
    # keyframes (~94 % of the frames) and the reference's dynamic skip rule (src/Cerebro.cpp:189-203)
    is_kf = np.array([rng.below(100) < 94 for _ in range(F)])
    est_ms = 58.0                                        # estimated_descriptor_compute_time_ms of a NetVLAD forward pass
    rows = []
    last_proc = None
    n_computed = 0
    for f in range(F):
        if not is_kf[f]:
            continue
        n_computed += 1
        incoming_ms = 1e9 if last_proc is None else (stamps_ns[f] - last_proc) / 1e6
        skip_frac = 1.0 - incoming_ms / est_ms
        last_proc = stamps_ns[f]
        if n_computed > 4 and rng.rand() / 32767.0 < skip_frac:
            continue
        rows.append(f)
    rows = np.array(rows, dtype=np.int64)
    ready_ns = stamps_ns[rows] + int(est_ms * 1e6)
    # the dot-product thread: ~10 Hz with stalls (plots, waitKey, the lock of the descriptor list)
    ticks = []
    t = stamps_ns[0] + 3 * 10**9
    end = stamps_ns[-1] + 2 * 10**9
    while t < end:
        ticks.append(int(np.searchsorted(ready_ns, t, side="right")))
        t += rng.pick([100, 100, 100, 100, 120, 150, 200, 350]) * 10**6
    ticks.append(len(rows))
    return dict(F=F, alpha=alpha, src=src, beta=beta, stamps_ns=stamps_ns, rows=rows, ticks=np.array(ticks, dtype=np.int64), revisits=revisits)


def descriptors(pl, seed: int, f64: bool):
    """All frames' descriptors (F x D float64 values; float32-representable unless f64) from the plan, through the C oracle."""
    F = pl["F"]
    out = np.empty((F, D), dtype=np.float64)
    chunk = 512
    for c0 in range(0, F, chunk):
        noise = oracle_lib.synth_rows(seed, range(c0, min(F, c0 + chunk)), D)
        for i in range(noise.shape[0]):
            f = c0 + i
            if f == 0:
                _ar1(None, noise[i], 0.0, not f64, out[0])
            elif pl["src"][f] >= 0:
                _ar1(out[pl["src"][f]], noise[i], float(pl["beta"][f]), not f64, out[f])
            else:
                _ar1(out[f - 1], noise[i], float(pl["alpha"][f]), not f64, out[f])
    return out


def make_run(variant: str = "mh01", seed: int = 20140624, f64: bool = False):
    """-> dict(db (N x D float64: the DB rows in arrival order), stamps [(sec, nsec)] per row, frame_idx per row, ticks, sha256)"""
    pl = plan(variant, seed)
    allf = descriptors(pl, seed, f64)
    db = np.ascontiguousarray(allf[pl["rows"]])
    st = pl["stamps_ns"][pl["rows"]]
    stamps = [(int(s // 10**9), int(s % 10**9)) for s in st]
    return dict(variant=variant, seed=seed, f64=f64, db=db, stamps=stamps, frame_idx=pl["rows"], ticks=pl["ticks"].tolist(),
                n_frames=pl["F"], n_revisits=len(pl["revisits"]), sha256=hashlib.sha256(db.tobytes()).hexdigest(),
                ticks_sha256=hashlib.sha256(pl["ticks"].tobytes()).hexdigest())


class _Tick(C.Structure):
    _fields_ = oracle_lib.OrcTickResult._fields_


def run_ticks(run, order: int, nthreads: int = 8, storage_f32: bool | None = None):
    """The whole tick schedule through orc_loop_tick_order.  order 1 = the reference's arithmetic (fp64 M, Eigen SSE2 GEMV order);
    order 0 = the device's fixed tree on the storage type the library would choose (float rows iff every value is float32).
    Returns (list of per-tick dicts for the ticks that scanned, found list in the reference's dump format)."""
    lib = oracle_lib.load()
    lib.orc_loop_tick_order.restype = None
    db64 = run["db"]
    if storage_f32 is None:
        storage_f32 = not run["f64"]
    if order == 0 and storage_f32:
        db = db64.astype(np.float32)
        assert np.array_equal(db.astype(np.float64), db64)
        elem = 4
    else:
        db, elem = db64, 8
    st = oracle_lib.OrcLoopState(0)
    p = oracle_lib.default_params()
    out = oracle_lib.OrcTickResult()
    gap = (C.c_double * 3)()
    ticks, found = [], []
    for l in run["ticks"]:
        lib.orc_loop_tick_order(C.byref(st), C.byref(p), db.ctypes.data_as(C.c_void_p), C.c_int32(elem), C.c_int32(D), C.c_int64(l),
                                C.c_int32(order), C.c_int32(nthreads), C.byref(out), gap)
        if out.status != 2:
            continue
        d = out.as_dict()
        d["l"] = l
        d["gap"] = list(gap)
        ticks.append(d)
        if out.found:
            a, b = run["stamps"][out.idx_curr], run["stamps"][out.idx_prev]
            found.append(dict(time_sec_a=a[0], time_nsec_a=a[1], time_sec_b=b[0], time_nsec_b=b[1],
                              time_double_a=a[0] + 1e-9 * a[1], time_double_b=b[0] + 1e-9 * b[1],
                              global_a=int(run["frame_idx"][out.idx_curr]), global_b=int(run["frame_idx"][out.idx_prev]), score=out.score))
    return ticks, found
