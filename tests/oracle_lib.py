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
    lib.orc_dot_tree_f32.restype = C.c_double
    lib.orc_dot_tree_f32.argtypes = [V, V, C.c_int32]
    lib.orc_dot_seq_f64.restype = C.c_double
    lib.orc_dot_seq_f64.argtypes = [V, V, C.c_int32]
    lib.orc_scan_topk_f32.restype = None
    lib.orc_scan_topk_f32.argtypes = [V, C.c_int64, C.c_int32, V, C.c_int32, C.c_int32, V, V]
    lib.orc_scan_topk_synth.restype = None
    lib.orc_scan_topk_synth.argtypes = [C.c_uint64, C.c_int64, C.c_int32, V, V, V, C.c_int64, V, C.c_int32, C.c_int32,
                                        V, V, C.c_int32]
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
def synth_rows(seed: int, rows, D: int, plants=()) -> np.ndarray:
    """rows: iterable of global row ids; plants: iterable of (dst, src, kind)."""
    lib = load()
    pm = {int(d): (int(s), int(k)) for d, s, k in plants}
    rows = list(rows)
    out = np.empty((len(rows), D), dtype=np.float32)
    for i, r in enumerate(rows):
        src, kind = pm.get(int(r), (-1, 0))
        lib.orc_synth_row_f32(seed, int(r), D, kind, src, _p(out[i]))
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


def scan_topk_synth(seed: int, k: int, D: int, queries: np.ndarray, K: int, plants=(), nthreads: int = 1):
    queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, D)
    nq = queries.shape[0]
    plants = sorted(plants)
    dst = np.array([p[0] for p in plants], dtype=np.int64)
    src = np.array([p[1] for p in plants], dtype=np.int64)
    kind = np.array([p[2] for p in plants], dtype=np.int32)
    sc = np.empty((nq, K), dtype=np.float64)
    ix = np.empty((nq, K), dtype=np.int64)
    load().orc_scan_topk_synth(seed, k, D, _p(dst), _p(src), _p(kind), len(plants), _p(queries), nq, K, _p(sc), _p(ix),
                               nthreads)
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


def ref_scan_f64_colmajor(M: np.ndarray, k: int, v, vm, vmm):
    """M: (cols, D) float64 array whose row i is column i of the reference's column-major M."""
    D = M.shape[1]
    u = np.empty(k); um = np.empty(k); umm = np.empty(k)
    maxv = np.empty(3); arg = np.empty(3, dtype=np.int64)
    load().orc_ref_scan_f64_colmajor(_p(M), D, k, _p(v), _p(vm), _p(vmm), _p(u), _p(um), _p(umm), _p(maxv), _p(arg))
    return maxv, arg, (u, um, umm)
