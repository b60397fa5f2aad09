"""bench.py's tick plan (positions, planted revisits, expected decisions) checked against the oracle on a small DB, so the
sanity asserts inside the timed benchmark are themselves pinned; plus the small host helpers."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402  (imports no torch / no GPU code at module level)
import oracle_lib  # noqa: E402


def test_planned_ticks_fire_exactly_where_the_oracle_fires():
    rows = 3000
    ls, plants, expect = bench.plan_ticks(rows, 13)
    assert ls[0] == rows + bench.LAG and all(b - a == 3 for a, b in zip(ls, ls[1:]))     # k = l - 50 >= rows scanned
    db = oracle_lib.synth_rows(bench.SEED, range(ls[-1]), bench.D, plants)
    orc = oracle_lib.LoopOracle(db)
    n_found = 0
    for l, e in zip(ls, expect):
        r = orc.tick(l)
        assert r["status"] == 2                                                            # CHIP_TICK_SCANNED
        if e is None:
            assert r["found"] == 0
        else:
            assert r["found"] == 1 and (r["idx_curr"], r["idx_prev"]) == e
            n_found += 1
    assert n_found == 4                                                                     # every 4th tick is a planted revisit


def test_long_runs_cycle_inside_the_tick_window():
    ls, plants, expect = bench.plan_ticks(100_000, 5000)
    assert len(ls) == bench.TICK_WINDOW == len(expect)
    assert ls[-1] - ls[0] < 4096                       # stays inside the sharded ctx's replicated ring (CHIP_RING_ROWS)


def test_small_helpers():
    assert bench.fmt_rows(1_000_000) == "1M" and bench.fmt_rows(125_000) == "125k" and bench.fmt_rows(1234) == "1234"
    assert 1 <= bench.usable_cpus() <= 4096


def test_size_legs_do_not_disturb_each_other():
    """bench.py's 10k / 100k legs tick over shorter prefixes of the SAME resident DB: their planted query rows lie inside the
    longer scans' prefixes, so the longer plans keep their revisited rows out of those windows.  Checked on the real sizes
    (no planted row is planted twice or used as a source) and, scaled down, against the oracle: every leg fires exactly where
    its own plan says with all the other legs' plants present in the DB."""
    LAG = bench.LAG

    def combined(main_rows, leg_rows, n_main, n_leg):
        windows = [(r, r + LAG + 3 * n_leg + 3) for r in leg_rows]
        plans = {main_rows: bench.plan_ticks(main_rows, n_main, avoid=windows)}
        for r in leg_rows:
            plans[r] = bench.plan_ticks(r, n_leg, avoid=[w for w in windows if w[0] < r])
        return plans, sorted(sum((p[1] for p in plans.values()), []))

    plans, allp = combined(1_000_000, (10_000, 100_000), 110, 260)
    dst = {p[0] for p in allp}
    assert len(dst) == len(allp)                                         # no row planted twice
    assert all(s not in dst for _, s, _ in allp)                         # every source is a plain synthetic row

    plans, allp = combined(9000, (3000, 5000), 16, 16)
    top = max(p[0][-1] for p in plans.values())
    db = oracle_lib.synth_rows(bench.SEED, range(top), 256, allp)
    for rows, (ls, _, expect) in plans.items():
        orc = oracle_lib.LoopOracle(db)
        for l, e in zip(ls, expect):
            r = orc.tick(l)
            assert r["status"] == 2
            assert (r["found"] == 0) if e is None else (r["found"] == 1 and (r["idx_curr"], r["idx_prev"]) == e), (rows, l, e, r)


def test_eigen_probe_reports_absence_or_a_number():
    """bench.py probes for a real Eigen at run time; without one it says so instead of assuming."""
    import bench
    r = bench.eigen_baseline(64, 0.05)
    assert r is None or (r[0] > 0 and r[1] >= 1)
