#!/usr/bin/env python3
"""Counts the fp64 operations per PnP hypothesis of bench.py's pnp-leg scene with the instrumented oracle build
(oracle/_build/liboracle_flops.so, -DORC_FLOP_COUNT) and writes profiles/pnp_flops.json -- the numerator of pnp.roofline
(bench.py reads the committed file: the oracle is not called from the GPU legs).  tests/test_oracle_flops.py checks the file
against a fresh count.   python scripts/count_pnp_flops.py [n_hypotheses]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_flops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
r = oracle_flops.bench_scene_flops_per_hypothesis(n)
algo = r["per_hypothesis"]
dense = algo - r["per_stage"][oracle_flops.STAGES[1]] + r["dense_lu_per_hypothesis"]
out = {"scene": "bench.py pnp_leg: make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242), sampler seed 4242, hypotheses 0.." + str(n - 1),
       "counted_by": "oracle/pnp_ransac.c built with -DORC_FLOP_COUNT (every +, -, *, / and sqrt = 1 operation; compares, fabs, exponent "
                     "scalings and integer work not counted); scripts/count_pnp_flops.py",
       "flops_per_hypothesis": algo,
       "flops_per_hypothesis_dense_elimination": dense,
       "dense_note": "the same with the Macaulay elimination's zero-multiplier rows NOT skipped (what the device's register-resident LU "
                     "executes: e - 0*u); the roofline is priced on the algorithmic count above",
       "per_stage": r["per_stage"], "model_fraction": r["model_fraction"], "n_hypotheses": n}
(ROOT / "profiles" / "pnp_flops.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
