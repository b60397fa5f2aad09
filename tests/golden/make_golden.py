"""Generates tests/golden/dot_scan_golden.json from the CPU oracle (oracle/dot_scan.c).

The reference holds no golden vectors for this path and cannot be built here, so these
vectors pin the ORACLE'S OWN behaviour (regression + GPU parity); inputs are regenerated from the seed by the
integer-domain generator, outputs (indices, scores as hex floats) are committed.
Run:  python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import oracle_lib  # noqa: E402
import scenarios  # noqa: E402

CASES = [dict(seed=20190412, N=500, D=256, n_loops=4, plant_seed=1),
         dict(seed=4242, N=260, D=4096, n_loops=2, plant_seed=2),
         dict(seed=77, N=900, D=1000, n_loops=5, plant_seed=3)]   # D not a multiple of 256


OUT = HERE / (sys.argv[1] if len(sys.argv) > 1 else "dot_scan_golden.json")


def refuse_to_overwrite(path):
    """tests/golden/FROZEN.json pins the v1 fixtures by checksum: they are golden vectors, not regression snapshots.  A kernel /
    oracle change that alters bits is judged AGAINST the frozen file (north-star tolerance); it never regenerates it."""
    if path.exists():
        raise SystemExit(f"{path.name} is frozen (tests/golden/FROZEN.json, tests/test_golden_frozen.py): refusing to overwrite. "
                         "Write additional cases to a NEW file (pass its name as argv[1]).")


def main():
    refuse_to_overwrite(OUT)
    out = {"generator": "oracle/dot_scan.c orc_synth_row_f32", "cases": []}
    for c in CASES:
        plants, loops, ties = scenarios.loop_plants(c["N"], c["n_loops"], c["plant_seed"])
        db = scenarios.build_db(c["seed"], c["N"], c["D"], plants)
        sched = scenarios.default_schedule(c["N"])
        orc = oracle_lib.LoopOracle(db)
        found = []
        for l in sched:
            r = orc.tick(l)
            if r["found"]:
                found.append([r["idx_curr"], r["idx_prev"], r["score"].hex()])
        assert len(found) >= len(loops), (found, loops)
        rows = [c["N"] - 1, c["N"] - 2, loops[0][1]]
        K = 8
        k = c["N"] - 50
        sc, ix = oracle_lib.scan_topk(db, k, db[rows], K)
        out["cases"].append(dict(seed=c["seed"], N=c["N"], D=c["D"], plants=[list(p) for p in plants], schedule=sched,
                                 loops=[list(x) for x in loops], ties=[list(t) for t in ties], found_loops=found,
                                 topk_rows=rows, topk_k=k, K=K, topk_idx=ix.tolist(),
                                 topk_scores_hex=[[float(x).hex() for x in row] for row in sc]))
    OUT.write_text(json.dumps(out, indent=1))
    print("wrote", HERE / "dot_scan_golden.json", [len(c["found_loops"]) for c in out["cases"]])


if __name__ == "__main__":
    main()
