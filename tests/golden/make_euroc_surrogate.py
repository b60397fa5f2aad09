#!/usr/bin/env python3
"""Writes tests/golden/euroc_surrogate_<variant>[_f64].json: the "recorded reference run" of the EuRoC-shaped surrogate
(tests/euroc_surrogate.py) -- the candidate list the REFERENCE's arithmetic selects (fp64 column-major M, Eigen 3.3 SSE2 row-major GEMV
order: oracle orc_loop_tick_order(order = 1), restating /root/reference/src/Cerebro.cpp:956-1100 with :1026-1028 in Eigen's order) in
the reference's dump format (loopcandidates_liverun.json, src/Cerebro.cpp:1127-1164, src/cerebro_node.cpp:769-770) -- plus what pins
the regenerated inputs (SHA-256 of the descriptor bytes and of the tick schedule) and the closeness statistics of the run.

  python tests/golden/make_euroc_surrogate.py [mh01] [mh01_f64] [mh01_05]        (refuses to overwrite an existing fixture)

The descriptors themselves are NOT committed (3067 x 4096 doubles = 100 MB; 11 k rows for the merged run): every machine regenerates
them bit for bit (C oracle + integer LCG) and checks the hash.  The real EuRoC data, the NetVLAD weights and a recorded run of the
reference are absent from the build image; this is the closest reference-shaped stand-in."""
import json
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import euroc_surrogate as E  # noqa: E402


def build(name: str):
    variant, f64 = (name[:-4], True) if name.endswith("_f64") else (name, False)
    nt = min(os.cpu_count() or 1, 32)
    run = E.make_run(variant, f64=f64)
    te, fe = E.run_ticks(run, 1, nt)                       # the reference's order: the recorded run
    tt, ft = E.run_ticks(run, 0, nt)                       # the device's order (what the GPU is bit-exact against)
    same = len(te) == len(tt) and all(a["argmax"] == b["argmax"] and a["found"] == b["found"] and a["idx_prev"] == b["idx_prev"] for a, b in zip(te, tt))
    gaps = np.array([a["gap"] for a in te])
    dev = max(abs(a["maxv"][q] - b["maxv"][q]) for a, b in zip(te, tt) for q in range(3))
    inc = np.diff([0] + run["ticks"])
    return dict(
        what="EuRoC-shaped surrogate run: recorded candidate list of the reference's arithmetic (Eigen 3.3 SSE2 GEMV order on fp64 M)",
        generator="tests/euroc_surrogate.py make_run(variant, seed, f64) + run_ticks(order=1); oracle/surrogate.c, oracle/dot_scan.c",
        variant=variant, seed=run["seed"], f64=f64, D=E.D, n_frames=run["n_frames"], n_rows=int(run["db"].shape[0]), n_ticks=len(run["ticks"]),
        n_ticks_scanned=len(te), n_revisits=run["n_revisits"], descriptors_sha256=run["sha256"], ticks_sha256=run["ticks_sha256"],
        tick_increment_histogram=np.bincount(inc)[:9].tolist(), max_tick_increment=int(inc.max()),
        min_top1_top2_gap=[float(x) for x in gaps.min(axis=0)], median_top1_top2_gap=[float(x) for x in np.median(gaps, axis=0)],
        ticks_with_gap_below_1e_4=int((gaps[:, 0] < 1e-4).sum()), ticks_above_threshold_rejected_by_locality=sum(1 for a in te if a["maxv"][0] > 0.85 and not a["found"]),
        ticks_within_0_005_of_threshold=sum(1 for a in te if abs(a["maxv"][0] - 0.85) < 0.005),
        tree_order_takes_the_same_decisions=bool(same), max_abs_score_deviation_tree_vs_eigen=float(dev),
        tree_order_scores_hex=[float(c["score"]).hex() for c in ft],
        loopcandidates_liverun=fe)


if __name__ == "__main__":
    names = sys.argv[1:] or ["mh01", "mh01_f64", "mh01_05"]
    for name in names:
        out = HERE / f"euroc_surrogate_{name}.json"
        if out.exists():
            print(f"{out.name} exists: fixtures are frozen once written (delete it by hand to regenerate)", file=sys.stderr)
            sys.exit(3)
        g = build(name)
        assert g["tree_order_takes_the_same_decisions"], "the device's summation order and the reference's disagree on this run"
        out.write_text(json.dumps(g, separators=(",", ":")).replace('},{"time_sec_a"', '},\n{"time_sec_a"') + "\n")
        print(name, {k: g[k] for k in ("n_rows", "n_ticks", "n_ticks_scanned", "min_top1_top2_gap", "ticks_with_gap_below_1e_4", "max_abs_score_deviation_tree_vs_eigen")},
              len(g["loopcandidates_liverun"]), "candidates")
