"""ctypes loader of the CPU oracle (oracle/*.c).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by cerebro_amd/."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_build" / "liboracle.so"


class OrcDotParams(C.Structure):
    _fields_ = [("locality", C.c_int32), ("thresh", C.c_double), ("lag", C.c_int32), ("min_new", C.c_int32),
                ("min_k", C.c_int32)]


class OrcTickResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("idx_curr", C.c_int64), ("idx_prev", C.c_int64),
                ("score", C.c_double), ("argmax", C.c_int64 * 3), ("maxv", C.c_double * 3)]

    def as_dict(self):
        return dict(status=self.status, found=self.found, idx_curr=self.idx_curr, idx_prev=self.idx_prev,
                    score=self.score, argmax=list(self.argmax), maxv=list(self.maxv))


class OrcLoopState(C.Structure):
    _fields_ = [("last_l", C.c_int64)]


_lib = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = list((ROOT / "oracle").glob("*.c")) + list((ROOT / "oracle").glob("*.h"))
    if not SO.exists() or any(s.stat().st_mtime > SO.stat().st_mtime for s in srcs):
        subprocess.run(["make", "oracle"], cwd=ROOT, check=True, capture_output=True)
    lib = C.CDLL(str(SO))
    V = C.c_void_p
    lib.orc_splitmix64.restype = C.c_uint64
    lib.orc_splitmix64.argtypes = [C.c_uint64]
    lib.orc_synth_i32.restype = C.c_int32
    lib.orc_synth_i32.argtypes = [C.c_uint64, C.c_int64, C.c_int32]
    lib.orc_synth_scale.restype = C.c_float
    lib.orc_synth_scale.argtypes = [C.c_int32]
    lib.orc_synth_scale_planted.restype = C.c_float
    lib.orc_synth_scale_planted.argtypes = [C.c_int32]
    lib.orc_synth_row_f32.restype = None
    lib.orc_synth_row_f32.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, V]
    lib.orc_synth_row_unit_f32.restype = None
    lib.orc_synth_row_unit_f32.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, V]
    lib.orc_dot_tree_f32.restype = C.c_double
    lib.orc_dot_tree_f32.argtypes = [V, V, C.c_int32]
    lib.orc_dot_tree_f64.restype = C.c_double
    lib.orc_dot_tree_f64.argtypes = [V, V, C.c_int32]
    lib.orc_scan_topk_f64.restype = None
    lib.orc_scan_topk_f64.argtypes = [V, C.c_int64, C.c_int32, V, C.c_int32, C.c_int32, V, V, C.c_int32]
    lib.orc_scores.restype = None
    lib.orc_scores.argtypes = [V, C.c_int32, C.c_int64, C.c_int32, V, V, C.c_int32]
    lib.orc_loop_tick_f64.restype = None
    lib.orc_loop_tick_f64.argtypes = [C.POINTER(OrcLoopState), C.POINTER(OrcDotParams), V, C.c_int32, C.c_int64,
                                      C.POINTER(OrcTickResult)]
    lib.orc_dot_seq_f64.restype = C.c_double
    lib.orc_dot_seq_f64.argtypes = [V, V, C.c_int32]
    lib.orc_scan_topk_f32.restype = None
    lib.orc_scan_topk_f32.argtypes = [V, C.c_int64, C.c_int32, V, C.c_int32, C.c_int32, V, V]
    lib.orc_dot_fmaf_f32.restype = C.c_float
    lib.orc_dot_fmaf_f32.argtypes = [V, V, C.c_int32]
    lib.orc_scan_topk_fmaf_f32.restype = None
    lib.orc_scan_topk_fmaf_f32.argtypes = [V, C.c_int64, C.c_int32, V, C.c_int32, C.c_int32, V, V]
    lib.orc_scan_topk_synth.restype = None
    lib.orc_scan_topk_synth.argtypes = [C.c_uint64, C.c_int64, C.c_int32, V, V, V, C.c_int64, V, C.c_int32, C.c_int32,
                                        V, V, C.c_int32]
    lib.orc_scan_topk_synth_unit.restype = None
    lib.orc_scan_topk_synth_unit.argtypes = lib.orc_scan_topk_synth.argtypes
    lib.orc_dot_params_default.restype = None
    lib.orc_dot_params_default.argtypes = [C.POINTER(OrcDotParams)]
    lib.orc_loop_tick_f32.restype = None
    lib.orc_loop_tick_f32.argtypes = [C.POINTER(OrcLoopState), C.POINTER(OrcDotParams), V, C.c_int32, C.c_int64,
                                      C.POINTER(OrcTickResult)]
    lib.orc_ref_scan_f64_colmajor.restype = None
    lib.orc_ref_scan_f64_colmajor.argtypes = [V, C.c_int32, C.c_int64, V, V, V, V, V, V, V, V]
    _lib = lib
    return lib


# ---------------------------------------------------------------- convenience wrappers
def synth_rows(seed: int, rows, D: int, plants=(), unit: bool = False) -> np.ndarray:
    """rows: iterable of global row ids; plants: iterable of (dst, src, kind); unit: the unit-L2 form of the generator."""
    lib = load()
    pm = {int(d): (int(s), int(k)) for d, s, k in plants}
    rows = list(rows)
    out = np.empty((len(rows), D), dtype=np.float32)
    gen = lib.orc_synth_row_unit_f32 if unit else lib.orc_synth_row_f32
    for i, r in enumerate(rows):
        src, kind = pm.get(int(r), (-1, 0))
        gen(seed, int(r), D, kind, src, _p(out[i]))
    return out


def dot_tree(q: np.ndarray, row: np.ndarray) -> float:
    q = np.ascontiguousarray(q, dtype=np.float32)
    row = np.ascontiguousarray(row, dtype=np.float32)
    return float(load().orc_dot_tree_f32(_p(q), _p(row), q.size))


def scan_topk(db: np.ndarray, k: int, queries: np.ndarray, K: int):
    db = np.ascontiguousarray(db, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, db.shape[1])
    nq = queries.shape[0]
    sc = np.empty((nq, K), dtype=np.float64)
    ix = np.empty((nq, K), dtype=np.int64)
    load().orc_scan_topk_f32(_p(db), k, db.shape[1], _p(queries), nq, K, _p(sc), _p(ix))
    return sc, ix


def scan_topk_fmaf(db: np.ndarray, k: int, queries: np.ndarray, K: int):
    """batched mode oracle: fp32 fmaf chain"""
    db = np.ascontiguousarray(db, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, db.shape[1])
    nq = queries.shape[0]
    sc = np.empty((nq, K), dtype=np.float64)
    ix = np.empty((nq, K), dtype=np.int64)
    load().orc_scan_topk_fmaf_f32(_p(db), k, db.shape[1], _p(queries), nq, K, _p(sc), _p(ix))
    return sc, ix


def scan_topk_synth(seed: int, k: int, D: int, queries: np.ndarray, K: int, plants=(), nthreads: int = 1, unit: bool = False):
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, D)
    nq = queries.shape[0]
    plants = sorted(plants)
    dst = np.array([p[0] for p in plants], dtype=np.int64)
    src = np.array([p[1] for p in plants], dtype=np.int64)
    kind = np.array([p[2] for p in plants], dtype=np.int32)
    sc = np.empty((nq, K), dtype=np.float64)
    ix = np.empty((nq, K), dtype=np.int64)
    fn = load().orc_scan_topk_synth_unit if unit else load().orc_scan_topk_synth
    fn(seed, k, D, _p(dst), _p(src), _p(kind), len(plants), _p(queries), nq, K, _p(sc), _p(ix), nthreads)
    return sc, ix


def default_params() -> OrcDotParams:
    p = OrcDotParams()
    load().orc_dot_params_default(C.byref(p))
    return p


class LoopOracle:
    """Replays Cerebro::descrip_N__dot__descrip_0_N ticks on a host fp32 DB."""

    def __init__(self, db: np.ndarray, params: OrcDotParams | None = None):
        self.db = np.ascontiguousarray(db, dtype=np.float32)
        self.state = OrcLoopState(0)
        self.params = params or default_params()

    def tick(self, l: int) -> dict:
        r = OrcTickResult()
        load().orc_loop_tick_f32(C.byref(self.state), C.byref(self.params), _p(self.db), self.db.shape[1], l, C.byref(r))
        return r.as_dict()


def loop_tick_order(db: np.ndarray, l: int, order: int, nthreads: int = 8) -> dict:
    """ONE tick at l from a fresh state (last_l = 0): order 0 = the device's fixed tree on float rows, order 1 = the reference's
    arithmetic (fp64 M, Eigen 3.3 SSE2 GEMV order).  db: float32-valued rows."""
    lib = load()
    lib.orc_loop_tick_order.restype = None
    if order == 0:
        m, elem = np.ascontiguousarray(db, dtype=np.float32), 4
    else:
        m, elem = np.ascontiguousarray(db, dtype=np.float64), 8
    st = OrcLoopState(0)
    p = default_params()
    out = OrcTickResult()
    gap = (C.c_double * 3)()
    lib.orc_loop_tick_order(C.byref(st), C.byref(p), m.ctypes.data_as(C.c_void_p), C.c_int32(elem), C.c_int32(m.shape[1]), C.c_int64(l),
                            C.c_int32(order), C.c_int32(nthreads), C.byref(out), gap)
    return out.as_dict()


class LoopOracle64:
    """The same ticks on a host fp64 DB (double-row storage mode): scores by orc_dot_tree_f64 (fma chains)."""

    def __init__(self, db: np.ndarray, params: OrcDotParams | None = None):
        self.db = np.ascontiguousarray(db, dtype=np.float64)
        self.state = OrcLoopState(0)
        self.params = params or default_params()

    def tick(self, l: int) -> dict:
        r = OrcTickResult()
        load().orc_loop_tick_f64(C.byref(self.state), C.byref(self.params), _p(self.db), self.db.shape[1], l, C.byref(r))
        return r.as_dict()


def dot_tree_f64(q: np.ndarray, row: np.ndarray) -> float:
    q = np.ascontiguousarray(q, dtype=np.float64)
    row = np.ascontiguousarray(row, dtype=np.float64)
    return float(load().orc_dot_tree_f64(_p(q), _p(row), q.size))


def scan_topk_f64(db: np.ndarray, k: int, queries: np.ndarray, K: int, nthreads: int = 1):
    db = np.ascontiguousarray(db, dtype=np.float64)
    queries = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, db.shape[1])
    nq = queries.shape[0]
    sc = np.empty((nq, K), dtype=np.float64)
    ix = np.empty((nq, K), dtype=np.int64)
    load().orc_scan_topk_f64(_p(db), k, db.shape[1], _p(queries), nq, K, _p(sc), _p(ix), nthreads)
    return sc, ix


def scores(db: np.ndarray, k: int, query: np.ndarray, nthreads: int = 1) -> np.ndarray:
    """u = v^T M[:, :k] in the device's summation order; db dtype float32 or float64 selects the path"""
    assert db.dtype in (np.float32, np.float64)
    db = np.ascontiguousarray(db)
    query = np.ascontiguousarray(query, dtype=db.dtype)
    u = np.empty(k, dtype=np.float64)
    load().orc_scores(_p(db), db.dtype.itemsize, k, db.shape[1], _p(query), _p(u), nthreads)
    return u


def ref_scan_f64_colmajor(M: np.ndarray, k: int, v, vm, vmm):
    """M: (cols, D) float64 array whose row i is column i of the reference's column-major M."""
    D = M.shape[1]
    u = np.empty(k); um = np.empty(k); umm = np.empty(k)
    maxv = np.empty(3); arg = np.empty(3, dtype=np.int64)
    load().orc_ref_scan_f64_colmajor(_p(M), D, k, _p(v), _p(vm), _p(vmm), _p(u), _p(um), _p(umm), _p(maxv), _p(arg))
    return maxv, arg, (u, um, umm)


def ref_scan_f64_colmajor_omp(M: np.ndarray, k: int, v, vm, vmm, nthreads: int, scratch=None):
    D = M.shape[1]
    u, um, umm = scratch if scratch is not None else (np.empty(k), np.empty(k), np.empty(k))
    maxv = np.empty(3); arg = np.empty(3, dtype=np.int64)
    load().orc_ref_scan_f64_colmajor_omp(_p(M), D, C.c_int64(k), _p(v), _p(vm), _p(vmm), _p(u), _p(um), _p(umm), _p(maxv), _p(arg),
                                         C.c_int32(nthreads))
    return maxv, arg, (u, um, umm)


def ref_scan_f64_eigen_order(M: np.ndarray, k: int, v, vm, vmm, packet: int = 2, fma: bool = False, nthreads: int = 1):
    """Cerebro.cpp:1026-1043 with Eigen 3.3's row-major GEMV summation order (packet 2, no FMA = the reference's SSE2 build)."""
    lib = load()
    lib.orc_ref_scan_f64_eigen_order.restype = None
    D = M.shape[1]
    M = np.ascontiguousarray(M, dtype=np.float64)
    v, vm, vmm = (np.ascontiguousarray(x, dtype=np.float64) for x in (v, vm, vmm))
    u = np.empty(k); um = np.empty(k); umm = np.empty(k)
    maxv = np.empty(3); arg = np.empty(3, dtype=np.int64)
    lib.orc_ref_scan_f64_eigen_order(_p(M), C.c_int32(D), C.c_int64(k), _p(v), _p(vm), _p(vmm), _p(u), _p(um), _p(umm), _p(maxv), _p(arg),
                                     C.c_int32(packet), C.c_int32(1 if fma else 0), C.c_int32(nthreads))
    return maxv, arg, (u, um, umm)


def ref_scan_f64_eigen_gemv3(M: np.ndarray, k: int, v, vm, vmm, nthreads: int = 1, scratch=None):
    """Cerebro.cpp:1026-1043 as the reference's SSE2 Release build runs it: three separate Eigen-shaped GEMVs (four rows at a time,
    one Packet2d accumulator each), maxCoeff, last-index argmax.  Entries are bit-identical to dot_eigen_gemv(packet=2)."""
    lib = load()
    lib.orc_ref_scan_f64_eigen_gemv3.restype = None
    D = M.shape[1]
    assert M.dtype == np.float64 and M.flags.c_contiguous
    v, vm, vmm = (np.ascontiguousarray(x, dtype=np.float64) for x in (v, vm, vmm))
    u, um, umm = scratch if scratch is not None else (np.empty(k), np.empty(k), np.empty(k))
    maxv = np.empty(3); arg = np.empty(3, dtype=np.int64)
    lib.orc_ref_scan_f64_eigen_gemv3(_p(M), C.c_int32(D), C.c_int64(k), _p(v), _p(vm), _p(vmm), _p(u), _p(um), _p(umm), _p(maxv), _p(arg),
                                     C.c_int32(nthreads))
    return maxv, arg, (u, um, umm)


def dot_eigen_gemv(v, col, packet: int = 2, fma: bool = False, aligned_start: int = 0) -> float:
    lib = load()
    lib.orc_dot_eigen_gemv_f64.restype = C.c_double
    v = np.ascontiguousarray(v, dtype=np.float64); col = np.ascontiguousarray(col, dtype=np.float64)
    return float(lib.orc_dot_eigen_gemv_f64(_p(v), _p(col), C.c_int32(v.size), C.c_int32(packet), C.c_int32(1 if fma else 0), C.c_int32(aligned_start)))


def tile_columns_omp(k: int, src: np.ndarray, nthreads: int) -> np.ndarray:
    """(k, D) float64 array made of repeated copies of `src`, first-touched by the threads that will scan it."""
    M = np.empty((k, src.shape[1]), dtype=np.float64)
    load().orc_tile_columns_omp(_p(M), src.shape[1], C.c_int64(k), _p(src), C.c_int64(src.shape[0]), C.c_int32(nthreads))
    return M


# ================================================================== PnP / RANSAC oracle bindings
class OrcRansacParams(C.Structure):
    _fields_ = [("error_thresh", C.c_double), ("min_inlier_ratio", C.c_double), ("max_iterations", C.c_int32),
                ("min_iterations", C.c_int32), ("use_mle", C.c_int32), ("sample_size", C.c_int32),
                ("failure_probability", C.c_double), ("seed", C.c_uint64), ("n_hypotheses", C.c_int32),
                ("sampler", C.c_int32)]


class OrcRansacSummary(C.Structure):
    _fields_ = [("n_iterations", C.c_int32), ("n_inliers", C.c_int32), ("best_hypothesis", C.c_int32),
                ("n_models", C.c_int32), ("best_cost", C.c_double)]


def _bind_pnp():
    lib = load()
    if getattr(lib, "_pnp_bound", False):
        return lib
    V = C.c_void_p
    lib.orc_rng_draw.restype = C.c_uint64
    lib.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_ransac_sample.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, V]
    lib.orc_dls_linear_form.argtypes = [C.c_uint64, C.c_int32, V]
    lib.orc_reproj_error.restype = C.c_double
    lib.orc_reproj_error.argtypes = [V, V, V]
    lib.orc_score_model.argtypes = [V, V, V, C.c_int32, C.c_double, C.c_int32, V, V, V]
    lib.orc_dls_monomial_positions.argtypes = [V]
    lib.orc_dls_cubics.argtypes = [V, V, C.c_int32, V, V]
    lib.orc_dls_action_matrix.restype = C.c_int
    lib.orc_dls_action_matrix.argtypes = [V, V, V]
    lib.orc_eig27_real.restype = C.c_int
    lib.orc_eig27_real.argtypes = [V, V, V]
    lib.orc_dls_pnp.restype = C.c_int
    lib.orc_dls_pnp.argtypes = [V, V, C.c_int32, V, V, V, C.c_int32]
    lib.orc_pnp_hypothesis.restype = C.c_int
    lib.orc_pnp_hypothesis.argtypes = [V, V, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, V, V]
    lib.orc_ransac_params_default.argtypes = [C.POINTER(OrcRansacParams)]
    lib.orc_ransac_max_iterations.restype = C.c_int32
    lib.orc_ransac_max_iterations.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32]
    lib.orc_pnp_ransac.restype = C.c_int
    lib.orc_pnp_ransac.argtypes = [V, V, C.c_int32, C.POINTER(OrcRansacParams), V, C.POINTER(C.c_float), V,
                                   C.POINTER(OrcRansacSummary)]
    lib._pnp_bound = True
    return lib


def ransac_params(**kw) -> OrcRansacParams:
    p = OrcRansacParams()
    _bind_pnp().orc_ransac_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ransac_sample_persistent(seed, H, N, S=15):
    """theia::RandomSampler's persistent permutation: the samples of hypotheses 0..H-1 -> [H, S]"""
    lib = _bind_pnp()
    lib.orc_ransac_sample_persistent.restype = None
    lib.orc_ransac_sample_persistent.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    out = np.empty((H, S), dtype=np.int32)
    lib.orc_ransac_sample_persistent(seed, H, N, S, _p(out))
    return out


def ransac_sample(seed, hyp, N, S=15):
    out = np.empty(S, dtype=np.int32)
    _bind_pnp().orc_ransac_sample(seed, hyp, N, S, _p(out))
    return out


def dls_linear_form(seed, hyp):
    u = np.empty(4)
    _bind_pnp().orc_dls_linear_form(seed, hyp, _p(u))
    return u


def dls_cubics(X, uv):
    X = np.ascontiguousarray(X, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
    T = np.empty(27); f = np.empty((3, 20))
    _bind_pnp().orc_dls_cubics(_p(X), _p(uv), X.shape[0], _p(T), _p(f))
    return T.reshape(3, 9), f


def dls_action_matrix(f, u):
    f = np.ascontiguousarray(f, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    S = np.empty((27, 27))
    rc = _bind_pnp().orc_dls_action_matrix(_p(f), _p(u), _p(S))
    return rc, S


def eig27_real(S):
    S = np.ascontiguousarray(S, dtype=np.float64)
    lam = np.empty(27); v4 = np.empty((27, 4))
    n = _bind_pnp().orc_eig27_real(_p(S), _p(lam), _p(v4))
    return n, lam[:max(n, 0)], v4[:max(n, 0)]


def dls_pnp(X, uv, u, max_out=8):
    X = np.ascontiguousarray(X, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    Rs = np.empty((max_out, 3, 3)); ts = np.empty((max_out, 3))
    n = _bind_pnp().orc_dls_pnp(_p(X), _p(uv), X.shape[0], _p(u), _p(Rs), _p(ts), max_out)
    return n, Rs[:max(0, min(n, max_out))], ts[:max(0, min(n, max_out))]


def pnp_hypothesis(X, uv, seed, hyp, S=15):
    X = np.ascontiguousarray(X, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
    T = np.empty(16); smp = np.empty(S, dtype=np.int32)
    ok = _bind_pnp().orc_pnp_hypothesis(_p(X), _p(uv), X.shape[0], seed, hyp, S, _p(T), _p(smp))
    return ok, T.reshape(4, 4).T.copy(), smp


def score_model(T, X, uv, thresh=0.03, use_mle=1):
    X = np.ascontiguousarray(X, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T.reshape(16))  # column-major
    cost = C.c_double(); nin = C.c_int32(); mask = np.zeros(X.shape[0], dtype=np.uint8)
    _bind_pnp().orc_score_model(_p(Tc), _p(X), _p(uv), X.shape[0], thresh, use_mle, C.byref(cost), C.byref(nin), _p(mask))
    return cost.value, nin.value, mask


def pnp_ransac(X, uv, params: OrcRansacParams | None = None):
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
    p = params or ransac_params()
    T = np.empty(16); conf = C.c_float(); mask = np.zeros(max(1, X.shape[0]), dtype=np.uint8); s = OrcRansacSummary()
    rc = _bind_pnp().orc_pnp_ransac(_p(X), _p(uv), X.shape[0], C.byref(p), _p(T), C.byref(conf), _p(mask), C.byref(s))
    return dict(status=rc, confidence=float(conf.value), T=T.reshape(4, 4).T.copy(), mask=mask[:X.shape[0]].copy(),
                summary=dict(n_iterations=s.n_iterations, n_inliers=s.n_inliers, best_hypothesis=s.best_hypothesis,
                             n_models=s.n_models, best_cost=s.best_cost))


def pnp_hypotheses_mt(X, uv, params: OrcRansacParams, H: int, nthreads: int):
    """Benchmark-mode hypotheses 0..H-1 on `nthreads` host threads; returns (winning hypothesis, number of models)."""
    lib = _bind_pnp()
    X = np.ascontiguousarray(X, dtype=np.float64); uv = np.ascontiguousarray(uv, dtype=np.float64)
    nm = C.c_int32()
    lib.orc_pnp_hypotheses_mt.restype = C.c_int32
    best = lib.orc_pnp_hypotheses_mt(_p(X), _p(uv), C.c_int32(X.shape[0]), C.byref(params), C.c_int32(H), C.c_int32(nthreads), C.byref(nm))
    return int(best), int(nm.value)


# ================================================================== Umeyama-ICP oracle bindings
def _bind_icp():
    lib = _bind_pnp()
    if getattr(lib, "_icp_bound", False):
        return lib
    V = C.c_void_p
    lib.orc_umeyama.restype = C.c_int
    lib.orc_umeyama.argtypes = [V, V, C.c_int32, V, V, C.POINTER(C.c_double)]
    lib.orc_icp_error.restype = C.c_double
    lib.orc_icp_error.argtypes = [V, V, V]
    lib.orc_icp_hypothesis.restype = C.c_int
    lib.orc_icp_hypothesis.argtypes = [V, V, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, V, C.POINTER(C.c_double)]
    lib.orc_icp_params_default.argtypes = [C.POINTER(OrcRansacParams)]
    lib.orc_icp_ransac.restype = C.c_int
    lib.orc_icp_ransac.argtypes = [V, V, C.c_int32, C.POINTER(OrcRansacParams), V, C.POINTER(C.c_float), V, C.POINTER(OrcRansacSummary)]
    lib._icp_bound = True
    return lib


def icp_params(**kw) -> OrcRansacParams:
    p = OrcRansacParams()
    _bind_icp().orc_icp_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def umeyama(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    R = np.empty(9); t = np.empty(3); s = C.c_double()
    rc = _bind_icp().orc_umeyama(_p(a), _p(b), a.shape[0], _p(R), _p(t), C.byref(s))
    return rc, R.reshape(3, 3), t, s.value


def icp_hypothesis(A, B, seed, hyp, S=10):
    A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
    T = np.empty(16); s = C.c_double()
    ok = _bind_icp().orc_icp_hypothesis(_p(A), _p(B), A.shape[0], seed, hyp, S, _p(T), C.byref(s))
    return ok, T.reshape(4, 4).T.copy(), s.value


def icp_ransac(A, B, params: OrcRansacParams | None = None):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1, 3)
    B = np.ascontiguousarray(B, dtype=np.float64).reshape(-1, 3)
    p = params or icp_params()
    T = np.empty(16); conf = C.c_float(); mask = np.zeros(max(1, A.shape[0]), dtype=np.uint8); s = OrcRansacSummary()
    rc = _bind_icp().orc_icp_ransac(_p(A), _p(B), A.shape[0], C.byref(p), _p(T), C.byref(conf), _p(mask), C.byref(s))
    return dict(status=rc, confidence=float(conf.value), T=T.reshape(4, 4).T.copy(), mask=mask[:A.shape[0]].copy(),
                summary=dict(n_iterations=s.n_iterations, n_inliers=s.n_inliers, best_hypothesis=s.best_hypothesis,
                             n_models=s.n_models, best_cost=s.best_cost))


# ================================================================== top-k candidate policies (oracle/policies.c)
class OrcPolicyLoop(C.Structure):
    _fields_ = [("idx_curr", C.c_int64), ("idx_prev", C.c_int64), ("score", C.c_double)]


class OrcNaiveState(C.Structure):
    _fields_ = [("last_l", C.c_int64), ("l_last_added", C.c_int64)]


class OrcCliqueState(C.Structure):
    _fields_ = [("last_l", C.c_int64), ("l_last_added", C.c_int64), ("n_retained", C.c_int32), ("pad_", C.c_int32),
                ("key", C.c_int64 * 64), ("cnt", C.c_int32 * 64)]


RND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class AnsiRand:
    """The ANSI C example rand() (seed 1) that cerebro_replay --policy installs as rand_source."""

    def __init__(self, seed=1):
        self.x = seed

    def __call__(self, *_):
        self.x = (self.x * 1103515245 + 12345) % (1 << 64)
        return (self.x // 65536) % 32768


class NaivePolicyOracle:
    def __init__(self, db):
        self.db = np.ascontiguousarray(db, dtype=np.float32)
        self.st = OrcNaiveState(0, 0)
        self.lib = load()
        self.lib.orc_faiss_naive_tick.restype = C.c_int32
        self.lib.orc_faiss_naive_tick.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(OrcNaiveState), C.POINTER(OrcPolicyLoop)]

    def tick(self, l):
        out = OrcPolicyLoop()
        n = self.lib.orc_faiss_naive_tick(_p(self.db), self.db.shape[1], l, C.byref(self.st), C.byref(out))
        return [(out.idx_curr, out.idx_prev, out.score)] if n else []


class CliquePolicyOracle:
    def __init__(self, db, rnd=None):
        self.db = np.ascontiguousarray(db, dtype=np.float32)
        self.st = OrcCliqueState()
        self.rnd = rnd or AnsiRand()
        self._cb = RND_FN(lambda _arg: self.rnd())
        self.lib = load()
        self.lib.orc_faiss_clique_tick.restype = C.c_int32
        self.lib.orc_faiss_clique_tick.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(OrcCliqueState), RND_FN, C.c_void_p,
                                                   C.POINTER(OrcPolicyLoop), C.c_int32]

    def tick(self, l):
        out = (OrcPolicyLoop * 32)()
        n = self.lib.orc_faiss_clique_tick(_p(self.db), self.db.shape[1], l, C.byref(self.st), self._cb, None, out, 32)
        assert n <= 32
        return [(out[i].idx_curr, out[i].idx_prev, out[i].score) for i in range(n)]
