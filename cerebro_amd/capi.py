"""ctypes binding of libcerebro_hip.so (include/cerebro_hip.h).

Plumbing only: every compute call goes through the C-ABI into the HIP kernels.  There is NO CPU
fallback -- if the shared library is missing or a call fails, a ChipError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
PRODUCT_LIB_PATH = _HERE / "lib" / "libcerebro_hip.so"
HOOKS_LIB_PATH = _HERE / "lib" / "hooks" / "libcerebro_hip.so"    # the TEST build (-DCHIP_TEST_HOOKS, `make testlibs`): fault injection; tests only


def _lib_path() -> Path:
    """CHIP_LIB=<path> loads another build of the library (same-box A/B runs, the degraded and the test builds) -- but only together with
    CHIP_ALLOW_LIB_OVERRIDE=1: a stray CHIP_LIB must not redirect a deployed process to some other .so."""
    override = os.environ.get("CHIP_LIB")
    if override:
        if os.environ.get("CHIP_ALLOW_LIB_OVERRIDE") == "1":
            return Path(override).resolve()
        sys.stderr.write(f"[cerebro_amd] CHIP_LIB={override} IGNORED: set CHIP_ALLOW_LIB_OVERRIDE=1 to load another build of the library\n")
    return PRODUCT_LIB_PATH


LIB_PATH = _lib_path()

CHIP_OK = 0
CHIP_ERR_INVALID_ARG = -1
CHIP_ERR_NO_DEVICE = -2
CHIP_ERR_HIP = -3
CHIP_ERR_OOM = -4
CHIP_ERR_NOT_F32 = -5
CHIP_ERR_NONFINITE = -6
CHIP_ERR_RANGE = -7
CHIP_ERR_UNSUPPORTED = -8
CHIP_ERR_TOO_FEW_POINTS = -9
CHIP_ERR_BUSY = -10
CHIP_ERR_COMM = -11
CHIP_ERR_SHARD_FAILED = -12
CHIP_ERR_GROUP_BROKEN = -13

CHIP_MAX_TOPK = 16
CHIP_MAX_NQ = 4
CHIP_DEFAULT_TOPK = 8
CHIP_RING_ROWS = 4096
CHIP_MAX_INFLIGHT = 64
CHIP_APPEND_ALLOW_ROUNDING = 1
CHIP_CREATE_STORE_F32 = 1
CHIP_CREATE_STORE_F64 = 2
CHIP_MULTI_EXCHANGE_COPY = 4
CHIP_COMM_ID_BYTES = 128
CHIP_EXCHANGE_NONE, CHIP_EXCHANGE_RCCL, CHIP_EXCHANGE_COPY = 0, 1, 2
CHIP_SCAN_FORM_ONE_ROW, CHIP_SCAN_FORM_ROWS = 1, 2
CHIP_SAMPLER_FRESH, CHIP_SAMPLER_THEIA_PERSISTENT = 0, 1

CHIP_TICK_SKIPPED, CHIP_TICK_TOO_SHORT, CHIP_TICK_SCANNED = 0, 1, 2


class ChipError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status} ({_strerror(status)}){' -- ' + detail if detail else ''}")


class DotParams(C.Structure):
    _fields_ = [("locality", C.c_int32), ("lag", C.c_int32), ("min_new", C.c_int32), ("min_k", C.c_int32),
                ("thresh", C.c_double)]


class TickResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("idx_curr", C.c_int64), ("idx_prev", C.c_int64),
                ("score", C.c_double), ("argmax", C.c_int64 * 3), ("maxv", C.c_double * 3)]

    def as_dict(self):
        return dict(status=self.status, found=self.found, idx_curr=self.idx_curr, idx_prev=self.idx_prev,
                    score=self.score, argmax=list(self.argmax), maxv=list(self.maxv))


class TopkEntry(C.Structure):
    _fields_ = [("score", C.c_double), ("idx", C.c_int64)]


class RansacParams(C.Structure):
    _fields_ = [("error_thresh", C.c_double), ("min_inlier_ratio", C.c_double), ("max_iterations", C.c_int32),
                ("min_iterations", C.c_int32), ("use_mle", C.c_int32), ("sample_size", C.c_int32),
                ("failure_probability", C.c_double), ("seed", C.c_uint64), ("n_hypotheses", C.c_int32),
                ("sampler", C.c_int32)]


class RansacSummary(C.Structure):
    _fields_ = [("n_iterations", C.c_int32), ("n_inliers", C.c_int32), ("best_hypothesis", C.c_int32),
                ("n_models", C.c_int32), ("best_cost", C.c_double)]


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("D", C.c_int32), ("device", C.c_int32), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32), ("n_cus", C.c_int32), ("rows_global", C.c_int64),
                ("rows_local", C.c_int64), ("capacity_local", C.c_int64), ("lossy_rows", C.c_int64),
                ("arch", C.c_char * 32), ("storage_bytes", C.c_int32), ("n_devices", C.c_int32), ("exchange", C.c_int32),
                ("comm_ranks", C.c_int32), ("comm_init_abandoned", C.c_int32), ("scan_forms", C.c_int32), ("test_hooks", C.c_int32)]


# every symbol include/cerebro_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGS = {
    "chip_strerror": (C.c_char_p, [C.c_int]),
    "chip_abi_version": (C.c_int, []),
    "chip_build_scan_forms": (C.c_int, []),
    "chip_build_test_hooks": (C.c_int, []),
    "chip_last_hip_error": (C.c_int, [_P, C.POINTER(C.c_char_p)]),
    "chip_last_comm_error": (C.c_int, [_P, C.POINTER(C.c_char_p)]),
    "chip_create": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "chip_create_ex": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]),
    "chip_create_multi": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int64, _P, C.c_int32, C.c_uint32]),
    "chip_comm_unique_id": (C.c_int, [_P]),
    "chip_comm_init_rank": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "chip_destroy": (None, [_P]),
    "chip_set_stream": (C.c_int, [_P, _P]),
    "chip_reset_stream": (C.c_int, [_P]),
    "chip_synchronize": (C.c_int, [_P]),
    "chip_db_append_f64": (C.c_int, [_P, _P, C.c_int64, C.c_uint32, C.POINTER(C.c_int64)]),
    "chip_db_append_f32": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "chip_db_size": (C.c_int64, [_P]),
    "chip_db_read_rows_f32": (C.c_int, [_P, _P, C.c_int64, _P]),
    "chip_db_read_rows_f64": (C.c_int, [_P, _P, C.c_int64, _P]),
    "chip_db_append_synthetic": (C.c_int, [_P, C.c_int64, C.c_uint64, _P, _P, _P, C.c_int64]),
    "chip_db_append_synthetic_unit": (C.c_int, [_P, C.c_int64, C.c_uint64, _P, _P, _P, C.c_int64]),
    "chip_query_rows": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P]),
    "chip_query_vectors_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P]),
    "chip_query_vectors_f64": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P]),
    "chip_query_scores": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "chip_query_batch_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, _P, _P]),
    "chip_dot_params_default": (None, [C.POINTER(DotParams)]),
    "chip_loop_tick": (C.c_int, [_P, C.c_int64, C.POINTER(DotParams), C.POINTER(TickResult)]),
    "chip_resident_pause": (C.c_int, [_P]),
    "chip_resident_resume": (C.c_int, [_P]),
    "chip_loop_tick_enqueue": (C.c_int, [_P, C.c_int64, C.POINTER(DotParams), C.c_int32]),
    "chip_loop_tick_collect": (C.c_int, [_P, C.c_int32, C.POINTER(TickResult)]),
    "chip_loop_last_l": (C.c_int64, [_P]),
    "chip_loop_reset": (None, [_P]),
    "chip_scan_local": (C.c_int, [_P, C.c_int64, C.POINTER(DotParams), C.c_int32, _P, C.POINTER(C.c_int32)]),
    "chip_merge_decide": (C.c_int, [_P, C.c_int64, C.POINTER(DotParams), _P, C.c_int32, C.c_int32, C.POINTER(TickResult)]),
    "chip_merge_decide_enqueue": (C.c_int, [_P, C.c_int64, C.POINTER(DotParams), _P, C.c_int32, C.c_int32, C.c_int32]),
    "chip_ransac_params_default": (None, [C.POINTER(RansacParams)]),
    "chip_pnp_ransac": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(RansacParams), _P, C.POINTER(C.c_float), _P,
                                  C.POINTER(RansacSummary)]),
    "chip_pnp_ransac_batch": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.POINTER(RansacParams), _P, _P, _P, _P, _P]),
    "chip_icp_params_default": (None, [C.POINTER(RansacParams)]),
    "chip_icp_ransac": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(RansacParams), _P, C.POINTER(C.c_float), _P,
                                  C.POINTER(RansacSummary)]),
    "chip_icp_ransac_enqueue": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(RansacParams)]),
    "chip_icp_ransac_collect": (C.c_int, [_P, _P, C.POINTER(C.c_float), _P, C.POINTER(RansacSummary)]),
    "chip_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "chip_profile_enable": (C.c_int, [_P, C.c_int32]),
    "chip_profile_reset": (C.c_int, [_P]),
    "chip_profile_scan": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load libcerebro_hip.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    # One HIP runtime per process.  The torch wheel bundles its own libamdhip64 (SONAME libamdhip64.so.7 but
    # NEEDED by file name), so a torch imported AFTER this library maps a second runtime that sees no GPUs.
    # When torch is installed, let it load its runtime first; this library then binds to the same one.
    if p.exists() and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: the system ROCm runtime is used
            pass
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: the HIP extension is not built. Run `make lib` (or __graft_entry__.build()). "
            "cerebro_amd has no CPU fallback.")
    lib = C.CDLL(str(p))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class use_hooks_library:
    """Context manager for tests/: Chip objects created inside it run on the TEST build of the library (lib/hooks/, fault-injection hooks
    compiled in); outside it -- and in every product path -- the product build, which has none.  Both builds may be mapped in one process
    (RTLD_LOCAL; each registers its own code objects)."""

    def __enter__(self):
        global _lib
        self.saved = _lib
        lib = load_library(HOOKS_LIB_PATH)
        if lib.chip_build_test_hooks() != 1:
            raise RuntimeError(f"{HOOKS_LIB_PATH} is not a -DCHIP_TEST_HOOKS build")
        _lib = lib
        return lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def declared_symbols():
    return sorted(_SIGS)


def _strerror(status: int) -> str:
    try:
        return load_library().chip_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "?"


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_dot_params() -> DotParams:
    p = DotParams()
    load_library().chip_dot_params_default(C.byref(p))
    return p


def default_ransac_params() -> RansacParams:
    p = RansacParams()
    load_library().chip_ransac_params_default(C.byref(p))
    return p


def default_icp_params() -> RansacParams:
    p = RansacParams()
    load_library().chip_icp_params_default(C.byref(p))
    return p


def comm_unique_id() -> bytes:
    """chip_comm_unique_id: 128 bytes rank 0 hands to every rank (any transport) before chip_comm_init_rank"""
    buf = C.create_string_buffer(CHIP_COMM_ID_BYTES)
    st = load_library().chip_comm_unique_id(buf)
    if st != CHIP_OK:
        raise ChipError(st, "chip_comm_unique_id")
    return buf.raw


class Chip:
    """Thin RAII wrapper over a chip_ctx.  storage: None = decided by the data (chip_create), "f32" / "f64" = chip_create_ex.
    devices=[...] makes ONE ctx over several GPUs of this process (chip_create_multi); copy_exchange forces the device-copy
    exchange instead of RCCL (implied when a device is named twice)."""

    def __init__(self, D: int, capacity_hint: int = 0, device: int = 0, shard_rank: int = 0, shard_count: int = 1,
                 storage: str | None = None, devices=None, copy_exchange: bool = False):
        self.lib = load_library()
        self.D = int(D)
        self.shard_rank, self.shard_count = shard_rank, shard_count
        flags = {None: 0, "f32": CHIP_CREATE_STORE_F32, "f64": CHIP_CREATE_STORE_F64}[storage]
        h = C.c_void_p()
        if devices is not None:
            dev = np.ascontiguousarray(devices, dtype=np.int32)
            st = self.lib.chip_create_multi(C.byref(h), D, capacity_hint, _ptr(dev), dev.size,
                                            flags | (CHIP_MULTI_EXCHANGE_COPY if copy_exchange else 0))
            where = "chip_create_multi"
        elif flags:
            st = self.lib.chip_create_ex(C.byref(h), D, capacity_hint, device, shard_rank, shard_count, flags)
            where = "chip_create_ex"
        else:
            st = self.lib.chip_create(C.byref(h), D, capacity_hint, device, shard_rank, shard_count)
            where = "chip_create"
        if st != CHIP_OK:
            raise ChipError(st, where)
        self.h = h

    def comm_init_rank(self, unique_id: bytes, n_ranks: int, rank: int):
        """attach an RCCL communicator to this sharded ctx (one process per GPU); afterwards loop_tick* work on it"""
        assert len(unique_id) == CHIP_COMM_ID_BYTES
        self._chk(self.lib.chip_comm_init_rank(self.h, C.c_char_p(unique_id), n_ranks, rank), "chip_comm_init_rank")

    # -- lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.lib.chip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st: int, where: str):
        if st != CHIP_OK:
            txt = C.c_char_p()
            if st == CHIP_ERR_COMM:
                r = self.lib.chip_last_comm_error(self.h, C.byref(txt))
                raise ChipError(st, where, f"ncclResult {r}: {txt.value.decode() if txt.value else ''}")
            hip = self.lib.chip_last_hip_error(self.h, C.byref(txt)) if st in (CHIP_ERR_HIP, CHIP_ERR_OOM) else 0
            raise ChipError(st, where, f"hipError {hip}: {txt.value.decode() if txt.value else ''}" if hip else "")

    def last_comm_error(self) -> str:
        """chip_last_comm_error: "<code>: <text>" of the last RCCL failure (or abandoned bootstrap) this ctx has seen"""
        txt = C.c_char_p()
        r = self.lib.chip_last_comm_error(self.h, C.byref(txt))
        return f"{r}: {txt.value.decode() if txt.value else ''}"

    def set_stream(self, stream_ptr: int | None):
        """None -> back to the ctx's private stream; an int is a hipStream_t handle (0 = HIP's null stream)."""
        if stream_ptr is None:
            self._chk(self.lib.chip_reset_stream(self.h), "chip_reset_stream")
        else:
            self._chk(self.lib.chip_set_stream(self.h, C.c_void_p(stream_ptr)), "chip_set_stream")

    def synchronize(self):
        self._chk(self.lib.chip_synchronize(self.h), "chip_synchronize")

    # -- DB
    def append_f64(self, desc: np.ndarray, allow_rounding: bool = False) -> int:
        desc = np.ascontiguousarray(desc, dtype=np.float64).reshape(-1, self.D)
        first = C.c_int64()
        self._chk(self.lib.chip_db_append_f64(self.h, _ptr(desc), desc.shape[0],
                                              CHIP_APPEND_ALLOW_ROUNDING if allow_rounding else 0, C.byref(first)),
                  "chip_db_append_f64")
        return first.value

    def append_f32(self, desc: np.ndarray) -> int:
        desc = np.ascontiguousarray(desc, dtype=np.float32).reshape(-1, self.D)
        first = C.c_int64()
        self._chk(self.lib.chip_db_append_f32(self.h, _ptr(desc), desc.shape[0], C.byref(first)), "chip_db_append_f32")
        return first.value

    def append_synthetic(self, n: int, seed: int, plants=(), unit: bool = False):
        """unit=True: the rows normalised to unit L2 norm (chip_db_append_synthetic_unit, SURVEY.md 8d's data)."""
        plants = sorted(plants)
        name = "chip_db_append_synthetic_unit" if unit else "chip_db_append_synthetic"
        fn = getattr(self.lib, name)
        if plants:
            dst = np.array([p[0] for p in plants], dtype=np.int64)
            src = np.array([p[1] for p in plants], dtype=np.int64)
            kind = np.array([p[2] for p in plants], dtype=np.int32)
            st = fn(self.h, n, seed, _ptr(dst), _ptr(src), _ptr(kind), len(plants))
        else:
            st = fn(self.h, n, seed, None, None, None, 0)
        self._chk(st, name)

    def size(self) -> int:
        return int(self.lib.chip_db_size(self.h))

    def read_rows(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.size, self.D), dtype=np.float32)
        self._chk(self.lib.chip_db_read_rows_f32(self.h, _ptr(rows), rows.size, _ptr(out)), "chip_db_read_rows_f32")
        return out

    def read_rows_f64(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((rows.size, self.D), dtype=np.float64)
        self._chk(self.lib.chip_db_read_rows_f64(self.h, _ptr(rows), rows.size, _ptr(out)), "chip_db_read_rows_f64")
        return out

    # -- queries
    def query_scores(self, k: int, row: int) -> np.ndarray:
        """the whole score vector u = v^T M[:, :k] of one query row (Cerebro.cpp:1026)"""
        u = np.full(max(k, 1), np.nan, dtype=np.float64)
        self._chk(self.lib.chip_query_scores(self.h, k, row, _ptr(u)), "chip_query_scores")
        return u[:k]

    def query_vectors_f64(self, k: int, q: np.ndarray, topk: int = CHIP_DEFAULT_TOPK):
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, self.D)
        nq = q.shape[0]
        sc = np.empty((nq, topk), dtype=np.float64)
        ix = np.empty((nq, topk), dtype=np.int64)
        self._chk(self.lib.chip_query_vectors_f64(self.h, k, _ptr(q), nq, topk, _ptr(sc), _ptr(ix)), "chip_query_vectors_f64")
        return sc, ix

    def query_rows(self, k: int, rows, topk: int = CHIP_DEFAULT_TOPK):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        nq = rows.size
        sc = np.empty((nq, topk), dtype=np.float64)
        ix = np.empty((nq, topk), dtype=np.int64)
        self._chk(self.lib.chip_query_rows(self.h, k, _ptr(rows), nq, topk, _ptr(sc), _ptr(ix)), "chip_query_rows")
        return sc, ix

    def query_vectors(self, k: int, q: np.ndarray, topk: int = CHIP_DEFAULT_TOPK):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.D)
        nq = q.shape[0]
        sc = np.empty((nq, topk), dtype=np.float64)
        ix = np.empty((nq, topk), dtype=np.int64)
        self._chk(self.lib.chip_query_vectors_f32(self.h, k, _ptr(q), nq, topk, _ptr(sc), _ptr(ix)),
                  "chip_query_vectors_f32")
        return sc, ix

    def query_batch(self, k: int, q: np.ndarray, topk: int = CHIP_DEFAULT_TOPK):
        """many-query fp32 MFMA mode: returns (scores float32 [Q, topk], idx int64 [Q, topk])"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.D)
        Q = q.shape[0]
        sc = np.empty((Q, topk), dtype=np.float32)
        ix = np.empty((Q, topk), dtype=np.int64)
        self._chk(self.lib.chip_query_batch_f32(self.h, k, _ptr(q), Q, topk, _ptr(sc), _ptr(ix)), "chip_query_batch_f32")
        return sc, ix

    # -- tick
    def loop_tick(self, l: int, params: DotParams | None = None) -> TickResult:
        p = params or default_dot_params()
        r = TickResult()
        self._chk(self.lib.chip_loop_tick(self.h, l, C.byref(p), C.byref(r)), "chip_loop_tick")
        return r

    def resident_pause(self):
        self._chk(self.lib.chip_resident_pause(self.h), "chip_resident_pause")

    def resident_resume(self):
        self._chk(self.lib.chip_resident_resume(self.h), "chip_resident_resume")

    def loop_tick_enqueue(self, l: int, slot: int, params: DotParams | None = None):
        p = params or default_dot_params()
        self._chk(self.lib.chip_loop_tick_enqueue(self.h, l, C.byref(p), slot), "chip_loop_tick_enqueue")

    def loop_tick_collect(self, slot: int) -> TickResult:
        r = TickResult()
        self._chk(self.lib.chip_loop_tick_collect(self.h, slot, C.byref(r)), "chip_loop_tick_collect")
        return r

    def loop_reset(self):
        self.lib.chip_loop_reset(self.h)

    def last_l(self) -> int:
        return int(self.lib.chip_loop_last_l(self.h))

    def scan_local(self, l: int, dev_out_ptr: int, topk: int = CHIP_DEFAULT_TOPK, params: DotParams | None = None) -> int:
        p = params or default_dot_params()
        status = C.c_int32()
        self._chk(self.lib.chip_scan_local(self.h, l, C.byref(p), topk, C.c_void_p(dev_out_ptr), C.byref(status)),
                  "chip_scan_local")
        return status.value

    def merge_decide(self, l: int, dev_gathered_ptr: int, n_lists: int, topk: int = CHIP_DEFAULT_TOPK,
                     params: DotParams | None = None) -> TickResult:
        p = params or default_dot_params()
        r = TickResult()
        self._chk(self.lib.chip_merge_decide(self.h, l, C.byref(p), C.c_void_p(dev_gathered_ptr), n_lists, topk,
                                             C.byref(r)), "chip_merge_decide")
        return r

    def merge_decide_enqueue(self, l: int, dev_gathered_ptr: int, n_lists: int, slot: int, topk: int = CHIP_DEFAULT_TOPK,
                             params: DotParams | None = None):
        p = params or default_dot_params()
        self._chk(self.lib.chip_merge_decide_enqueue(self.h, l, C.byref(p), C.c_void_p(dev_gathered_ptr), n_lists, topk, slot),
                  "chip_merge_decide_enqueue")

    # -- PnP
    def pnp_ransac(self, X: np.ndarray, uv: np.ndarray, params: RansacParams | None = None):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        N = X.shape[0]
        assert uv.shape[0] == N
        p = params or default_ransac_params()
        T = np.empty(16, dtype=np.float64)
        conf = C.c_float()
        mask = np.zeros(max(N, 1), dtype=np.uint8)
        summ = RansacSummary()
        st = self.lib.chip_pnp_ransac(self.h, _ptr(X), _ptr(uv), N, C.byref(p), _ptr(T), C.byref(conf), _ptr(mask),
                                      C.byref(summ))
        if st == CHIP_ERR_TOO_FEW_POINTS:
            return dict(status=st, confidence=-1.0, T=None, mask=None, summary=None)
        self._chk(st, "chip_pnp_ransac")
        return dict(status=st, confidence=float(conf.value), T=T.reshape(4, 4).T.copy(), mask=mask[:N].copy(),
                    summary=dict(n_iterations=summ.n_iterations, n_inliers=summ.n_inliers,
                                 best_hypothesis=summ.best_hypothesis, n_models=summ.n_models,
                                 best_cost=summ.best_cost))

    def pnp_ransac_batch(self, problems, params: RansacParams | None = None, seeds=None):
        """problems: list of (X (N_i,3), uv (N_i,2)).  One pair of launches for the whole list; entry i equals
        pnp_ransac(X_i, uv_i) run with seed seeds[i] (default: params.seed for all)."""
        P = len(problems)
        Xs = [np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3) for X, _ in problems]
        uvs = [np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2) for _, uv in problems]
        Ns = np.array([x.shape[0] for x in Xs], dtype=np.int32)
        p = params or default_ransac_params()
        T = np.empty((P, 16), dtype=np.float64)
        conf = np.zeros(P, dtype=np.float32)
        masks = [np.zeros(max(int(n), 1), dtype=np.uint8) for n in Ns]
        summ = (RansacSummary * max(P, 1))()
        Xp = (C.c_void_p * max(P, 1))(*[x.ctypes.data for x in Xs])
        uvp = (C.c_void_p * max(P, 1))(*[u.ctypes.data for u in uvs])
        mp = (C.c_void_p * max(P, 1))(*[m.ctypes.data for m in masks])
        sd = None if seeds is None else np.asarray(seeds, dtype=np.uint64)
        st = self.lib.chip_pnp_ransac_batch(self.h, P, Xp, uvp, _ptr(Ns), C.byref(p), None if sd is None else _ptr(sd),
                                            _ptr(T), _ptr(conf), mp, summ)
        self._chk(st, "chip_pnp_ransac_batch")
        return [dict(status=st, confidence=float(conf[i]), T=T[i].reshape(4, 4).T.copy(), mask=masks[i][:Ns[i]].copy(),
                     summary=dict(n_iterations=summ[i].n_iterations, n_inliers=summ[i].n_inliers,
                                  best_hypothesis=summ[i].best_hypothesis, n_models=summ[i].n_models,
                                  best_cost=summ[i].best_cost)) for i in range(P)]

    def icp_ransac_enqueue(self, A: np.ndarray, B: np.ndarray, params: RansacParams | None = None):
        """start an ICP estimation on the ctx's ICP stream (returns at once); fetch it with icp_ransac_collect(N)"""
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1, 3)
        B = np.ascontiguousarray(B, dtype=np.float64).reshape(-1, 3)
        p = params or default_icp_params()
        self._chk(self.lib.chip_icp_ransac_enqueue(self.h, _ptr(A), _ptr(B), A.shape[0], C.byref(p)), "chip_icp_ransac_enqueue")
        return A.shape[0]

    def icp_ransac_collect(self, N: int):
        T = np.empty(16, dtype=np.float64)
        conf = C.c_float()
        mask = np.zeros(max(N, 1), dtype=np.uint8)
        summ = RansacSummary()
        self._chk(self.lib.chip_icp_ransac_collect(self.h, _ptr(T), C.byref(conf), _ptr(mask), C.byref(summ)), "chip_icp_ransac_collect")
        return dict(status=0, confidence=float(conf.value), T=T.reshape(4, 4).T.copy(), mask=mask[:N].copy(),
                    summary=dict(n_iterations=summ.n_iterations, n_inliers=summ.n_inliers, best_hypothesis=summ.best_hypothesis,
                                 n_models=summ.n_models, best_cost=summ.best_cost))

    def icp_ransac(self, A: np.ndarray, B: np.ndarray, params: RansacParams | None = None):
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1, 3)
        B = np.ascontiguousarray(B, dtype=np.float64).reshape(-1, 3)
        N = A.shape[0]
        assert B.shape[0] == N
        p = params or default_icp_params()
        T = np.empty(16, dtype=np.float64)
        conf = C.c_float()
        mask = np.zeros(max(N, 1), dtype=np.uint8)
        summ = RansacSummary()
        st = self.lib.chip_icp_ransac(self.h, _ptr(A), _ptr(B), N, C.byref(p), _ptr(T), C.byref(conf), _ptr(mask), C.byref(summ))
        if st == CHIP_ERR_TOO_FEW_POINTS:
            return dict(status=st, confidence=-1.0, T=None, mask=None, summary=None)
        self._chk(st, "chip_icp_ransac")
        return dict(status=st, confidence=float(conf.value), T=T.reshape(4, 4).T.copy(), mask=mask[:N].copy(),
                    summary=dict(n_iterations=summ.n_iterations, n_inliers=summ.n_inliers,
                                 best_hypothesis=summ.best_hypothesis, n_models=summ.n_models, best_cost=summ.best_cost))

    # -- introspection / profiling
    def info(self) -> dict:
        i = Info()
        self._chk(self.lib.chip_get_info(self.h, C.byref(i)), "chip_get_info")
        return {k: (getattr(i, k).decode() if k == "arch" else getattr(i, k)) for k, _ in Info._fields_}

    def profile_enable(self, on: bool = True):
        self._chk(self.lib.chip_profile_enable(self.h, 1 if on else 0), "chip_profile_enable")

    def profile_reset(self):
        self._chk(self.lib.chip_profile_reset(self.h), "chip_profile_reset")

    def profile_scan(self):
        ms, n, b, span = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.chip_profile_scan(self.h, C.byref(ms), C.byref(n), C.byref(b), C.byref(span)), "chip_profile_scan")
        return ms.value, n.value, b.value, span.value
