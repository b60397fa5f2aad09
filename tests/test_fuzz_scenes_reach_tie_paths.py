"""Which inputs reach the tie paths of the GPU's pivot search (pnp_build_solve's factor wave: 64-bit compares when two candidates agree in
the high word of their magnitude, the smallest-logical-index rule -- with the reference's row swaps replayed from the pivot history -- on
exact ties)?  Generic data never does, so the GPU fuzz carries symmetric scenes for it (tests/pnp_fuzz_scenes.py, kinds 8-11).  This test
makes that claim checkable without a GPU: the oracle built with -DORC_LU_TIE_STATS (oracle/_build/liboracle_stats.so, test infrastructure)
counts the ties per pivot column while it runs the same scenes with the same seeds as tests/test_fuzz_gpu.py."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_build" / "liboracle_stats.so"

CHILD = r"""
import ctypes as C, sys, pathlib
import numpy as np
sys.path.insert(0, sys.argv[1] + "/tests"); sys.path.insert(0, sys.argv[1])
import oracle_lib as O
O.SO = pathlib.Path(sys.argv[2])
from pnp_fuzz_scenes import scene
lib = O._bind_pnp()
st = (C.c_long * 4).in_dll(lib, "orc_lu_tie_stats")
rng = np.random.default_rng(7)
per_kind = {}
for i in range(64):                                   # the 64 scenes tests/test_fuzz_gpu.py runs, same seeds
    X, uv = scene(i, rng)
    before = list(st)
    for H in (0, 60):
        O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=H, seed=5000 + i))
    rng.normal(0, 0.01, X.shape)                       # (the ICP part of the fuzz draws from the same generator)
    k = per_kind.setdefault(i % 12, [0, 0, 0])
    for j in range(3):
        k[j] += st[j] - before[j]
print(dict(per_kind), st[3])
"""


@pytest.mark.skipif(not SO.exists(), reason="oracle/_build/liboracle_stats.so not built (make oracle)")
def test_fuzz_scenes_reach_tie_paths():
    r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), str(SO)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    per_kind, last_step = eval(r.stdout.strip().rsplit(" ", 1)[0]), int(r.stdout.strip().rsplit(" ", 1)[1])
    generic = [per_kind[k] for k in range(8)]
    assert sum(c[0] for c in generic) > 100_000                          # pivot columns seen
    assert sum(c[1] for c in generic) == 0 and sum(c[2] for c in generic) == 0   # kinds 0-7: not one tie of either sort
    symmetric = [per_kind[k] for k in (8, 9, 10, 11)]
    assert all(c[2] > 0 for c in symmetric)                              # every symmetric kind produces exact ties ...
    assert sum(c[2] for c in symmetric) > 500 and sum(c[1] for c in symmetric) > 500
    assert last_step >= 64                                               # ... also past step 64 (second half of the pivot history)
