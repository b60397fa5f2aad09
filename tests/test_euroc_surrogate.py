"""BASELINE configs 1 and 5 on reference-SHAPED data (the EuRoC bags, the NetVLAD weights and a recorded run of the reference are
absent from this image): tests/euroc_surrogate.py regenerates a 4096-D trajectory-correlated descriptor stream with revisits, the
reference's keyframe-skip rule and an irregular tick schedule; tests/golden/euroc_surrogate_*.json hold its "recorded reference run"
-- the candidate list selected by the reference's arithmetic (fp64 M, Eigen 3.3 SSE2 GEMV order) in loopcandidates_liverun.json
format.  CPU: the inputs regenerate bit for bit, the Eigen-order run IS the committed one, and the device's fixed-tree order takes
the same decision at every tick although best and second-best scores come as close as 7e-7.  GPU: cerebro_replay over the same
stream and `cerebro_replay --compare` against the recorded run -- on one device, on an 8-way row-sharded group, float and double rows."""
import json
import os
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import euroc_surrogate as E

ROOT = Path(__file__).resolve().parent.parent
GOLD = Path(__file__).resolve().parent / "golden"
LIB = ROOT / "cerebro_amd" / "lib"
NT = min(os.cpu_count() or 1, 32)

_runs = {}


def get_run(name):
    if name not in _runs:
        variant, f64 = (name[:-4], True) if name.endswith("_f64") else (name, False)
        _runs[name] = E.make_run(variant, f64=f64)
    return _runs[name]


def strip_scores(cands):
    return [{k: v for k, v in c.items() if k not in ("score", "time_double_a", "time_double_b")} for c in cands]


@pytest.mark.parametrize("name", ["mh01", "mh01_f64"])
def test_surrogate_regenerates_and_both_orders_select_the_recorded_run(name):
    g = json.loads((GOLD / f"euroc_surrogate_{name}.json").read_text())
    run = get_run(name)
    assert run["sha256"] == g["descriptors_sha256"] and run["ticks_sha256"] == g["ticks_sha256"]      # inputs regenerate bit for bit
    assert run["db"].shape == (g["n_rows"], 4096) and 2900 <= g["n_rows"] <= 3200                      # ~3k keyframes (config 1)
    assert np.allclose(np.linalg.norm(run["db"], axis=1), 1.0, atol=1e-6)
    is_f32 = np.array_equal(run["db"].astype(np.float32).astype(np.float64), run["db"])
    assert is_f32 == (not run["f64"])
    inc = np.diff([0] + run["ticks"])
    assert inc.min() == 0 and 6 <= np.sort(inc)[-2] <= 8 and (inc[1:] < 3).mean() > 0.3                 # irregular arrival, many idle ticks
    # the reference's arithmetic reproduces the committed recorded run exactly (scores included)
    te, fe = E.run_ticks(run, 1, NT)
    assert fe == g["loopcandidates_liverun"] and len(fe) > 100
    # the device's summation order: same argmax triple, same accept / reject, same reported index at EVERY tick
    tt, ft = E.run_ticks(run, 0, NT)
    assert len(tt) == len(te) == g["n_ticks_scanned"]
    for a, b in zip(te, tt):
        assert a["argmax"] == b["argmax"] and a["found"] == b["found"] and a["idx_prev"] == b["idx_prev"], (a, b)
    assert strip_scores(ft) == strip_scores(fe)
    assert [float(c["score"]).hex() for c in ft] == g["tree_order_scores_hex"]
    gaps = np.array([a["gap"] for a in te])
    dev = max(abs(a["maxv"][q] - b["maxv"][q]) for a, b in zip(te, tt) for q in range(3))
    print(f"\n[{name}] {len(te)} scanned ticks, {len(fe)} candidates; smallest top-1 / top-2 gap {gaps.min():.3e} "
          f"({(gaps[:, 0] < 1e-4).sum()} ticks below 1e-4); max |score_tree - score_eigen| = {dev:.2e}")
    assert gaps.min() < 1e-5 and gaps.min() > 100 * dev            # close argmax, yet far above the summation noise
    assert dev < 1e-15 * 64                                       # 1e-15 * sqrt(D)
    assert g["ticks_above_threshold_rejected_by_locality"] >= 5 and g["ticks_within_0_005_of_threshold"] >= 10


def test_merged_fixture_is_consistent():
    """config 5 (MH-01..05 merged): the committed recorded run is checked for shape here; regenerating and replaying it is the GPU test
    (the CPU oracle needs ~10 core-minutes for the 11 k-row run)."""
    g = json.loads((GOLD / "euroc_surrogate_mh01_05.json").read_text())
    assert g["n_frames"] == 3682 + 3040 + 2700 + 2033 + 2273 and 10_000 < g["n_rows"] < 13_000
    assert g["tree_order_takes_the_same_decisions"] is True and len(g["loopcandidates_liverun"]) > 500
    assert min(g["min_top1_top2_gap"]) < 1e-6 and g["max_abs_score_deviation_tree_vs_eigen"] < 6.4e-14
    for c in g["loopcandidates_liverun"]:
        assert c["time_sec_a"] > c["time_sec_b"] and c["score"] > 0.85 and c["global_a"] > c["global_b"]


def write_stream(path, run, frames="FRMS"):
    """+ the frames of the run, so that the replay's dump carries the reference's global_a / global_b (index into the map of ALL
    camera frames, Cerebro.cpp:1142-1143): as the stamp of every frame (FRMS: what DataManager's data_map holds) or as a row -> frame
    index table (FIDX)."""
    db = run["db"]
    N, D = db.shape
    with open(path, "wb") as f:
        f.write(b"CRBR" + struct.pack("<IIQQ", 1, D, N, len(run["ticks"])))
        f.write(np.asarray(run["stamps"], dtype=np.uint32).tobytes())
        f.write(db.tobytes())
        f.write(np.asarray(run["ticks"], dtype=np.int64).tobytes())
        if frames == "FRMS":
            ns = E.plan(run["variant"], run["seed"])["stamps_ns"]
            st = np.stack([ns // 10**9, ns % 10**9], axis=1).astype(np.uint32)
            f.write(b"FRMS" + struct.pack("<Q", len(st)) + st.tobytes())
        elif frames == "FIDX":
            f.write(b"FIDX" + struct.pack("<Q", N) + np.asarray(run["frame_idx"], dtype=np.int64).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("name,devices", [("mh01", None), ("mh01", "0,0,0,0,0,0,0,0"), ("mh01_f64", None), ("mh01_f64", "0,0,0,0,0,0,0,0"),
                                          ("mh01_05", None), ("mh01_05", "0,0,0,0,0,0,0,0")])
def test_gpu_replay_selects_the_recorded_reference_run(tmp_path, name, devices):
    """cerebro_replay (the C++ host mirror over the C ABI) fed the surrogate stream, then `cerebro_replay --compare` against the
    recorded Eigen-order run: identical selection, scores within the summation noise; and the GPU's own scores are, bit for bit, the
    fixed-tree oracle's (committed as hex).  devices = 8 x device 0: BASELINE config 4 / 5's 8-way row shard on the 1-GPU box."""
    g = json.loads((GOLD / f"euroc_surrogate_{name}.json").read_text())
    run = get_run(name)
    assert run["sha256"] == g["descriptors_sha256"] and run["ticks_sha256"] == g["ticks_sha256"]
    write_stream(tmp_path / "s.bin", run, "FIDX" if devices else "FRMS")
    cmd = [str(LIB / "cerebro_replay")] + (["--devices", devices] if devices else []) + [str(tmp_path / "s.bin"), str(tmp_path / "ours.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    (tmp_path / "loopcandidates_liverun.json").write_text(json.dumps(g["loopcandidates_liverun"], indent=4))     # nlohmann dump(4) shape
    r = subprocess.run([str(LIB / "cerebro_replay"), "--compare", str(tmp_path / "loopcandidates_liverun.json"), str(tmp_path / "ours.json")],
                       capture_output=True, text=True)
    rep = json.loads(r.stdout)
    assert r.returncode == 0 and rep["identical_selection"] and rep["n_reference"] == rep["n_candidate"] == len(g["loopcandidates_liverun"]), rep
    # the strict form: EVERY field of the recorded dump, global_a / global_b (indices into the map of all frames) included
    assert rep["identical_dump"] is True and rep["global_index_mismatches"] == 0 and rep["global_index_compared"] == len(g["loopcandidates_liverun"]), rep
    assert rep["max_abs_score_diff"] < 6.4e-14
    ours = json.loads((tmp_path / "ours.json").read_text())
    assert [float(c["score"]).hex() for c in ours] == g["tree_order_scores_hex"]
