"""Counted fp64 operations of the oracle's PnP solver (oracle/_build/liboracle_flops.so, -DORC_FLOP_COUNT): test infrastructure and
bench.py's pricing of the PnP kernels' fp64-vector roofline -- never on the product path."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_build" / "liboracle_flops.so"
STAGES = ("cost matrix + cubics", "Macaulay elimination (LU, zero multipliers skipped)", "back-substitution + action matrix",
          "Hessenberg reduction + accumulation", "Francis QR", "real eigenvectors + back-transform", "pose + cheirality",
          "reprojection scoring (all N points)")


def count_hypotheses(X, uv, seed: int, hyps, S: int = 15, thresh: float = 0.03):
    """Per-stage operation counts summed over the given hypothesis indices of one problem (sample -> DLS -> eigen -> pose -> score over
    all N points when the hypothesis yields a model).  Returns (counts[8], dense_lu, n_models)."""
    lib = C.CDLL(str(SO))
    lib.orc_pnp_hypothesis.restype = C.c_int
    lib.orc_pnp_hypothesis.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.orc_score_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_flop_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    X = np.ascontiguousarray(X, dtype=np.float64)
    uv = np.ascontiguousarray(uv, dtype=np.float64)
    N = X.shape[0]
    lib.orc_flop_counts(None, None, 1)
    T = np.zeros(16)
    cost, nin = C.c_double(), C.c_int32()
    n_models = 0
    for h in hyps:
        if lib.orc_pnp_hypothesis(X.ctypes.data, uv.ctypes.data, N, seed, int(h), S, T.ctypes.data, None):
            n_models += 1
            lib.orc_score_model(T.ctypes.data, X.ctypes.data, uv.ctypes.data, N, thresh, 1, C.byref(cost), C.byref(nin), None)
    out = (C.c_long * 8)()
    dense = C.c_long()
    lib.orc_flop_counts(out, C.byref(dense), 1)
    return np.array(list(out), dtype=np.int64), int(dense.value), n_models


def bench_scene_flops_per_hypothesis(n_hyp: int = 200):
    """The bench scene of bench.py's pnp leg (512 correspondences, 30 % outliers, seed 4242): mean operations per hypothesis."""
    import sys
    sys.path.insert(0, str(ROOT))
    from cerebro_amd.synth import make_scene
    X, uv, _, _ = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    counts, dense, n_models = count_hypotheses(X, uv, 4242, range(n_hyp))
    return {"per_hypothesis": float(counts.sum()) / n_hyp, "per_stage": {s: float(c) / n_hyp for s, c in zip(STAGES, counts)},
            "dense_lu_per_hypothesis": dense / n_hyp, "model_fraction": n_models / n_hyp, "n_hypotheses": n_hyp}
