"""The v1 golden fixtures are frozen (a fixture that is regenerated when the kernel is optimised is a
regression snapshot, not a golden vector).  tests/golden/FROZEN.json holds their checksums; the generators refuse to
overwrite them; the current oracle must still reproduce them at the north-star bar."""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

import np_mirror_pnp as M
import oracle_lib as O

GOLD = Path(__file__).parent / "golden"


def test_fixture_checksums():
    fz = json.loads((GOLD / "FROZEN.json").read_text())
    assert fz["version"] == 1 and set(fz["sha256"]) == {"dot_scan_golden.json", "pnp_golden.json"}
    for name, want in fz["sha256"].items():
        assert hashlib.sha256((GOLD / name).read_bytes()).hexdigest() == want, f"{name} was modified: v1 fixtures are frozen"


def test_generators_refuse_to_overwrite():
    for gen in ("make_golden.py", "make_golden_pnp.py"):
        r = subprocess.run([sys.executable, str(GOLD / gen)], capture_output=True, text=True)
        assert r.returncode != 0 and "frozen" in (r.stderr + r.stdout)


def test_current_oracle_meets_the_frozen_pnp_fixture_at_the_north_star_bar():
    """Whatever the oracle does internally today, against the FROZEN file: hypothesis indices, inlier masks and iteration
    counts bit-exact, pose within 1e-4 relative Frobenius (BASELINE.json north_star)."""
    g = json.loads((GOLD / "pnp_golden.json").read_text())
    X, uv = np.array(g["X"]), np.array(g["uv"])
    for case in g["cases"]:
        r = O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=case["n_hypotheses"], seed=case["seed"]))
        s = r["summary"]
        assert (s["best_hypothesis"], s["n_iterations"], s["n_models"], s["n_inliers"]) == \
            (case["best_hypothesis"], case["n_iterations"], case["n_models"], case["n_inliers"])
        assert np.packbits(r["mask"]).tobytes().hex() == case["mask_hex"]
        T = np.array([float.fromhex(x) for x in case["T_colmajor_hex"]]).reshape(4, 4).T
        assert np.linalg.norm(r["T"] - T) <= 1e-4 * np.linalg.norm(T)
