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


def _synthetic_record(n_gpus=1):
    """A record shaped like the one main() hands to finalize_record at N = 1, every leg present, distinct values everywhere."""
    leg = lambda x: {"value": 1e4 + x, "ms_per_step": 0.1 + x, "sync_tick_us": 40.0 + x, "roofline": {"frac_kernel": 0.5 + x / 100, "frac_step": 0.9,
                                                                                                      "frac_kernel_actual_bytes": 0.8 + x / 100}}   # noqa: E731
    stat = lambda x: {"mean_us": x + 1.0, "p50_us": float(x), "min_us": x - 1.0}   # noqa: E731
    return {
        "metric": "m", "value": 422.0, "unit": "loop-queries/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 2.37,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "w", "db_rows": 1_000_000, "D": 4096, "queries_per_tick": 3, "topk": 8, "storage": "s", "sharding": "single GPU", "exchange": "none"},
        "roofline": {"bound": "hbm", "achieved": 6900.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.8625, "traffic": 1.6e10, "traffic_source": "file",
                     "kernel": "db_scan_topk", "avg_kernel_ms": 2.37, "launches": 20, "measured": "hipEvents", "algorithmic_bytes_per_launch": 1.6384e10,
                     "pure_read_ceiling": {"value": 7090.0}},
        "sizes": {"10k": leg(1), "29k": leg(2), "100k": leg(3), "1M": {"value": 422.0}},
        "shapes": {"8192x29k": leg(4), "f64_1M": leg(5)},
        "paced_10hz": {"first": {"launched": stat(52), "resident": stat(36)}, "after_contexts": {"launched": stat(70), "resident": stat(37)},
                       "spin": {"launched": stat(47), "resident": stat(35.4)}},
        "resident_tick": {"10k": {"launched": stat(40), "resident": stat(35)}, "29k": {"launched": stat(95), "resident": stat(90)}},
        "pnp": {"value": 1.4e6, "ms_per_call_1000_hyp": 0.71, "batch8_hypotheses_per_s": 2.5e6, "reference_mode_ms_per_call": 0.46,
                "reference_mode_pair_ms_per_call": 0.47, "cpu_baseline": {"value": 7000.0},
                "roofline": {"frac": 0.0156, "frac_batch8": 0.028, "flops_per_hypothesis": 864600.8, "valu_busy": {"pnp_build_solve": 0.31, "pnp_eig_score": 0.63},
                             "valu_insts_per_hypothesis": {"pnp_build_solve": 37556.0}}},
        "icp": {"value": 7e7, "three_way_pose_ms": 0.5}, "batch": {"value": 15900.0, "roofline": {"frac": 0.84}},
    }


def test_record_layout_keeps_both_halves_of_the_metric_inside_what_the_driver_keeps():
    """VERDICT r5 missing 1: the driver's record kept the first 24 `config` keys and the scalars of `roofline`; the PnP half of
    BASELINE's metric and the size legs' isolated-launch fractions were appended after that and dropped.  finalize_record puts
    them (a) first in `config` after `workload`, with `config` held to <= 24 keys, (b) inside `roofline` as scalars, before any text."""
    rec = bench.finalize_record(_synthetic_record())
    json_rt = __import__("json").loads(__import__("json").dumps(rec))          # what the driver parses: key order survives
    cfg, roof = json_rt["config"], json_rt["roofline"]
    assert len(cfg) <= bench.CONFIG_KEY_CAP == 24
    want = ["pnp_hyp_per_s", "pnp_batch8_hypotheses_per_s", "pnp_reference_mode_ms_per_call", "pnp_reference_mode_pair_ms_per_call",
            "pnp_roofline_frac", "pnp_roofline_frac_batch8", "size_10k_roofline_frac_kernel", "size_29k_roofline_frac_kernel",
            "size_100k_roofline_frac_kernel", "size_8192x29k_roofline_frac_kernel", "size_f64_1M_roofline_frac_kernel",
            "size_f64_1M_roofline_frac_actual_bytes", "size_10k_sync_tick_10hz_launched_us", "size_10k_sync_tick_10hz_resident_us"]
    assert list(cfg)[0] == "workload" and list(cfg)[1:1 + len(want)] == want
    assert [k for k, _ in bench.HEADLINE_SCALARS][:len(want)] == want
    # the same scalars inside `roofline`: after the contract's six keys and the dominant kernel's own scalars, before every string
    rk = list(roof)
    assert rk[:6] == ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    first_text = min(i for i, k in enumerate(rk) if i >= 6 and isinstance(roof[k], str))
    for k in want + ["size_10k_sync_tick_10hz_launched_after_contexts_us", "pnp_valu_insts_per_hypothesis_build_solve", "batch256_roofline_frac"]:
        assert k in roof and rk.index(k) < first_text and isinstance(roof[k], (int, float)), k
    assert not any(isinstance(v, dict) for v in roof.values())                # nested context moved out (the driver drops dicts)
    assert json_rt["roofline_context"]["pure_read_ceiling"]["value"] == 7090.0
    # values are the legs' own
    assert cfg["pnp_hyp_per_s"] == 1.4e6 and cfg["pnp_roofline_frac_batch8"] == 0.028 and roof["pnp_reference_mode_pair_ms_per_call"] == 0.47
    assert abs(cfg["size_29k_roofline_frac_kernel"] - 0.52) < 1e-12 and abs(cfg["size_f64_1M_roofline_frac_actual_bytes"] - 0.85) < 1e-12
    assert cfg["size_10k_sync_tick_10hz_launched_us"] == 52.0 and roof["size_10k_sync_tick_10hz_launched_after_contexts_us"] == 70.0
    assert roof["size_10k_sync_tick_10hz_spin_launched_us"] == 47.0 and roof["size_10k_sync_tick_10hz_spin_resident_us"] == 35.4   # (round 6: the pause spun through)
    # the contract's own numbers are untouched
    assert roof["frac"] == 0.8625 and roof["achieved"] == 6900.0 and roof["kernel"] == "db_scan_topk" and json_rt["value"] == 422.0


def test_record_layout_without_legs_and_with_too_many_config_keys():
    rec = _synthetic_record(n_gpus=8)
    for k in ("sizes", "shapes", "paced_10hz", "resident_tick", "pnp", "icp", "batch"):
        del rec[k]
    rec["config"].update({f"k{i}": i for i in range(30)})
    out = bench.finalize_record(rec)
    assert len(out["config"]) == 24 and list(out["config"])[:3] == ["workload", "db_rows", "D"]
    assert len(out["config_more"]) == 8 + 30 - 24 and "k29" in out["config_more"]
    assert "pnp_hyp_per_s" not in out["roofline"] and out["roofline"]["frac"] == 0.8625
