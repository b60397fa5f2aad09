"""Generates tests/golden/pnp_golden.json from the CPU oracle (oracle/pnp_ransac.c) on the SURVEY 8d scene.
PARITY UNPINNED w.r.t. the live reference (Theia absent, RNG unseeded there): these vectors pin the oracle's own
definition and serve as the GPU parity target.  Run: python tests/golden/make_golden_pnp.py"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import np_mirror_pnp as M  # noqa: E402
import oracle_lib as O  # noqa: E402


OUT = HERE / (sys.argv[1] if len(sys.argv) > 1 else "pnp_golden.json")


def refuse_to_overwrite(path):
    """tests/golden/FROZEN.json pins the v1 fixtures by checksum: they are golden vectors, not regression snapshots.  A kernel /
    oracle change that alters bits is judged AGAINST the frozen file (north-star tolerance); it never regenerates it."""
    if path.exists():
        raise SystemExit(f"{path.name} is frozen (tests/golden/FROZEN.json, tests/test_golden_frozen.py): refusing to overwrite. "
                         "Write additional cases to a NEW file (pass its name as argv[1]).")


def main():
    refuse_to_overwrite(OUT)
    X, uv, T, inl = M.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
    out = dict(scene="np_mirror_pnp.make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)",
               X=X.tolist(), uv=uv.tolist(), T_true=T.tolist(), cases=[])
    for nh, seed in [(0, 4242), (0, 1), (64, 2), (1000, 4242)]:
        r = O.pnp_ransac(X, uv, O.ransac_params(n_hypotheses=nh, seed=seed))
        s = r["summary"]
        out["cases"].append(dict(n_hypotheses=nh, seed=seed, best_hypothesis=s["best_hypothesis"],
                                 n_iterations=s["n_iterations"], n_models=s["n_models"], n_inliers=s["n_inliers"],
                                 best_cost_hex=float(s["best_cost"]).hex(), confidence=r["confidence"],
                                 mask_hex=np.packbits(r["mask"]).tobytes().hex(),
                                 T_colmajor_hex=[float(x).hex() for x in r["T"].T.reshape(16)]))
        print(nh, seed, s)
    OUT.write_text(json.dumps(out))
    print("wrote pnp_golden.json", OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
