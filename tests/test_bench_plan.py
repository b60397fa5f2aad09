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
