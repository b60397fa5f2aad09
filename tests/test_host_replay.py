"""The C++ host mirror (cerebro_amd/host) and the replay harness: build/link checks on CPU, end-to-end replay on GPU."""
import json
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import scenarios

pytestmark = pytest.mark.needs_hip_build   # uses libcerebro_hip.so / the host binaries (conftest skips these without hipcc)
ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "cerebro_amd" / "lib"


def write_stream(path, db, stamps, ticks):
    N, D = db.shape
    with open(path, "wb") as f:
        f.write(b"CRBR" + struct.pack("<IIQQ", 1, D, N, len(ticks)))
        f.write(np.asarray(stamps, dtype=np.uint32).tobytes())
        f.write(db.astype(np.float64).tobytes())
        f.write(np.asarray(ticks, dtype=np.int64).tobytes())


def test_host_library_and_replay_are_built_and_link_against_the_c_abi():
    assert (LIB / "libcerebro_host.so").exists() and (LIB / "cerebro_replay").exists()
    out = subprocess.run(["ldd", str(LIB / "cerebro_replay")], capture_output=True, text=True).stdout
    assert "libcerebro_host.so" in out and "libcerebro_hip.so" in out and "not found" not in out
    syms = subprocess.run(["nm", "-DC", str(LIB / "libcerebro_host.so")], capture_output=True, text=True).stdout
    for name in ("cerebro_hip::Cerebro::foundLoops_count", "cerebro_hip::Cerebro::foundLoops_i", "cerebro_hip::Cerebro::foundLoops_as_JSON",
                 "cerebro_hip::Cerebro::wholeImageComputedList_size", "cerebro_hip::Cerebro::wholeImageComputedList_at",
                 "cerebro_hip::Cerebro::descrip_N__dot__descrip_0_N_once", "cerebro_hip::StaticTheiaPoseCompute::PNP",
                 "cerebro_hip::make_loop_edge"):
        assert name in syms, name
    # the host library computes nothing itself: its only undefined chip_* symbols are the C ABI
    und = [l.split()[-1] for l in syms.splitlines() if " U chip_" in l]
    assert set(und) >= {"chip_create", "chip_loop_tick", "chip_db_append_f64", "chip_pnp_ransac"}


def test_replay_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    db = scenarios.build_db(1, 60, 64, [])
    write_stream(tmp_path / "s.bin", db, [(i, 0) for i in range(60)], [60])
    r = subprocess.run([str(LIB / "cerebro_replay"), str(tmp_path / "s.bin"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert r.returncode == 3 and "chip_create failed" in r.stderr          # no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0", "0,0,0"])
def test_replay_matches_oracle(tmp_path, devices):
    """devices: the C++ host class over chip_create_multi -- "0" = a one-device group over an RCCL communicator, "0,0,0" = the
    three-way sharded code path on one device (device-copy exchange); the reference's call sites read the same either way."""
    D, N = 512, 1300
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=21)
    db = scenarios.build_db(77, N, D, plants)
    stamps = [(1403636579 + i // 20, (i % 20) * 50_000_000) for i in range(N)]     # 20 Hz keyframes
    ticks = scenarios.default_schedule(N)
    write_stream(tmp_path / "s.bin", db, stamps, ticks)
    r = subprocess.run([str(LIB / "cerebro_replay")] + (["--devices", devices] if devices else []) + [str(tmp_path / "s.bin"), str(tmp_path / "o.json")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads((tmp_path / "o.json").read_text())
    orc = oracle_lib.LoopOracle(db)
    want = []
    for l in ticks:
        o = orc.tick(l)
        if o["found"]:
            a, b = stamps[o["idx_curr"]], stamps[o["idx_prev"]]
            want.append(dict(time_sec_a=a[0], time_nsec_a=a[1], time_sec_b=b[0], time_nsec_b=b[1],
                             global_a=o["idx_curr"], global_b=o["idx_prev"], score=o["score"]))
    assert len(got) == len(want) >= len(loops)
    for g, w in zip(got, want):
        for k, v in w.items():
            assert g[k] == v, (k, g, w)            # score round-trips exactly through %.17g
        assert g["time_double_a"] == pytest.approx(w["time_sec_a"] + 1e-9 * w["time_nsec_a"])
    # the one-command check for whoever holds a recorded run of the reference: its dump vs ours
    (tmp_path / "liverun.json").write_text(json.dumps([dict(w, time_double_a=0.0, time_double_b=0.0, global_a=0, global_b=0) for w in want], indent=4))
    # (this stream carries no frame list: the dump's global_a / global_b are DB rows, so only the SELECTION is compared)
    r = subprocess.run([str(LIB / "cerebro_replay"), "--compare-selection", str(tmp_path / "liverun.json"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert r.returncode == 0 and json.loads(r.stdout)["identical_selection"], r.stdout + r.stderr
    # ... and with the frames known (every 20 Hz keyframe is every 2nd camera frame of a 40 Hz stream: data_map index = 2 x row) the
    # dump carries the reference's global_a / global_b (Cerebro.cpp:1142-1143) and the strict form passes, by either trailer
    for trailer in ("FRMS", "FIDX"):
        with open(tmp_path / "s2.bin", "wb") as f:
            f.write((tmp_path / "s.bin").read_bytes())
            if trailer == "FRMS":
                frames = []
                for (sec, nsec) in stamps:
                    frames += [(sec, nsec), (sec, nsec + 25_000_000)]
                f.write(b"FRMS" + struct.pack("<Q", len(frames)) + np.asarray(frames, dtype=np.uint32).tobytes())
            else:
                f.write(b"FIDX" + struct.pack("<Q", N) + (2 * np.arange(N, dtype=np.int64)).tobytes())
        r = subprocess.run([str(LIB / "cerebro_replay")] + (["--devices", devices] if devices else []) + [str(tmp_path / "s2.bin"), str(tmp_path / "o2.json")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        (tmp_path / "liverun2.json").write_text(json.dumps([dict(w, time_double_a=0.0, time_double_b=0.0, global_a=2 * w["global_a"], global_b=2 * w["global_b"])
                                                            for w in want], indent=4))
        r = subprocess.run([str(LIB / "cerebro_replay"), "--compare", str(tmp_path / "liverun2.json"), str(tmp_path / "o2.json")], capture_output=True, text=True)
        j = json.loads(r.stdout)
        assert r.returncode == 0 and j["identical_dump"] and j["global_index_mismatches"] == 0 and j["global_index_compared"] == len(want), (trailer, j)
        r = subprocess.run([str(LIB / "cerebro_replay"), "--compare", str(tmp_path / "liverun2.json"), str(tmp_path / "o.json")], capture_output=True, text=True)
        assert r.returncode == 1 and json.loads(r.stdout)["global_index_mismatches"] > 0        # rows are not data_map indices


def test_compare_recorded_reference_run(tmp_path):
    """cerebro_replay --compare: diff of a recorded loopcandidates_liverun.json (the reference's own dump, nlohmann dump(4),
    src/cerebro_node.cpp:769-770) against this harness's output.  No GPU: pure host logic."""
    rng = np.random.default_rng(5)
    ref = []
    for i in range(40):
        sa, na = 1403636579 + 3 * i, int(rng.integers(0, 10**9))
        sb, nb = 1403636000 + i, int(rng.integers(0, 10**9))
        ref.append({"time_sec_a": sa, "time_nsec_a": na, "time_sec_b": sb, "time_nsec_b": nb, "time_double_a": sa + 1e-9 * na,
                    "time_double_b": sb + 1e-9 * nb, "global_a": 4000 + 7 * i, "global_b": 100 + i, "score": float(rng.uniform(0.86, 0.99))})

    def run(a, b, mode="--compare-selection"):
        (tmp_path / "ref.json").write_text(a if isinstance(a, str) else json.dumps(a, indent=4))
        (tmp_path / "ours.json").write_text(b if isinstance(b, str) else json.dumps(b, separators=(",", ":")))
        r = subprocess.run([str(LIB / "cerebro_replay"), mode, str(tmp_path / "ref.json"), str(tmp_path / "ours.json")], capture_output=True, text=True)
        return r.returncode, (json.loads(r.stdout) if r.stdout.strip() else r.stderr)

    ours = [dict(c, global_a=c["global_a"] // 7, global_b=c["global_b"] - 100, score=c["score"] * (1 + 2e-16)) for c in ref]   # other index space, last-bit scores
    rc, j = run(ref, ours)
    assert rc == 0 and j["identical_selection"] and j["n_reference"] == j["n_candidate"] == j["matched_prefix"] == 40
    assert 0 < j["max_abs_score_diff"] < 1e-15
    # the strict form (every field of a genuine loopcandidates_liverun.json): another index space does not pass, the same one does
    rc, j = run(ref, ours, "--compare")
    assert rc == 1 and j["identical_selection"] and not j["identical_dump"] and j["global_index_mismatches"] == 40
    same_space = [dict(c, score=c["score"] * (1 + 2e-16)) for c in ref]
    rc, j = run(ref, same_space, "--compare")
    assert rc == 0 and j["identical_dump"] and j["global_index_compared"] == 40 and j["global_index_mismatches"] == 0
    same_space[3]["global_b"] += 1
    rc, j = run(ref, same_space, "--compare")
    assert rc == 1 and j["global_index_mismatches"] == 1
    bad = [dict(c) for c in ours]
    bad[17]["time_nsec_b"] += 50_000_000                       # a neighbouring keyframe was selected
    rc, j = run(ref, bad)
    assert rc == 1 and not j["identical_selection"] and j["matched_prefix"] == 17 and j["first_divergence"]["index"] == 17
    assert j["first_divergence"]["reference"]["time_nsec_b"] + 50_000_000 == j["first_divergence"]["candidate"]["time_nsec_b"]
    rc, j = run(ref, ours[:30])                                 # candidates missing at the end
    assert rc == 1 and j["matched_prefix"] == 30 and j["first_divergence"]["candidate"] is None
    rc, j = run(ref, ours[:5] + ours[6:])                       # one candidate missing in the middle
    assert rc == 1 and j["first_divergence"]["index"] == 5
    rc, j = run("null", "[]")                                   # nlohmann dumps a never-pushed-to json as null
    assert rc == 0 and j["n_reference"] == 0
    rc, j = run(ref, '[{"time_sec_a": 1}]')
    assert rc == 6 and "score" in j


# ------------------------------------------------------------------ N1: state.json cold start
def write_state_json(path, db, stamps_nsec, has_desc=None, digits=15):
    """Same structure as DataManager::saveStateToDisk (DataManager.cpp:1098-1215): nlohmann dump(4); descriptor text is
    Eigen IOFormat(FullPrecision, DontAlignCols, ", ", "\\n") = one value per line with 15 significant digits."""
    nodes = []
    for i, ns in enumerate(stamps_nsec):
        node = {"stampNSec": int(ns), "stamp_relative": i * 0.05, "seq": i, "isKeyFrame": True,
                "getNumberOfSuccessfullyTrackedFeatures": 77, "isPoseAvailable": True,
                "w_T_c": {"rows": 4, "cols": 4, "stampNSec": int(ns), "data": "1, 0, 0, 0\n0, 1, 0, 0\n0, 0, 1, 0\n0, 0, 0, 1",
                          "data_pretty": ":YPR(deg)=(0,0,0)  \"quoted\" \\ back"}}
        avail = has_desc is None or has_desc[i]
        if avail:
            node["wholeImageDescriptor"] = {"rows": db.shape[1], "cols": 1,
                                            "data": "\n".join(f"{float(v):.{digits}g}" for v in db[i])}
        node["isWholeImageDescriptorAvailable"] = bool(avail)
        nodes.append(node)
    doc = {"DataNodes": nodes, "ImageDataManager": {"stash": [{"a": [1, 2, {"b": None}]}], "n": 3}}
    Path(path).write_text(json.dumps(doc, indent=4))
    return doc


def test_state_json_parser_matches_python(tmp_path):
    D, N = 64, 40
    db = scenarios.build_db(3, N, D, [])
    stamps = [1403636579_000000000 + i * 50_000_000 for i in range(N)]
    has = [i % 5 != 2 for i in range(N)]                               # some nodes are not keyframes / have no descriptor
    doc = write_state_json(tmp_path / "state.json", db, stamps, has)
    r = subprocess.run([str(LIB / "cerebro_replay"), "--parse-only", str(tmp_path / "state.json"), str(tmp_path / "o.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "o.bin").read_bytes()
    Dg, n = struct.unpack_from("<IQ", raw, 0)
    assert Dg == D and n == sum(has)
    got_stamps = np.frombuffer(raw, dtype=np.uint64, count=n, offset=12)
    got = np.frombuffer(raw, dtype=np.float64, count=n * D, offset=12 + 8 * n).reshape(n, D)
    want_stamps = [s for s, h in zip(stamps, has) if h]
    want = np.array([[float(x) for x in nd["wholeImageDescriptor"]["data"].split("\n")] for nd in doc["DataNodes"] if "wholeImageDescriptor" in nd])
    assert list(got_stamps) == want_stamps
    assert got.tobytes() == want.tobytes()                              # strtod == Python float(): both correctly rounded
    # 15 significant digits do not round-trip a double, but rounding to float32 recovers the original descriptor
    orig = db[[i for i in range(N) if has[i]]]
    assert not np.array_equal(got, orig.astype(np.float64))
    assert np.array_equal(got.astype(np.float32), orig)
    # malformed input is an error, not a crash
    (tmp_path / "bad.json").write_text('{"DataNodes": [ {"stampNSec": 5, "wholeImageDescriptor": {"rows": 3, "cols": 1, "data": "1\\n2"}} ]}')
    r = subprocess.run([str(LIB / "cerebro_replay"), "--parse-only", str(tmp_path / "bad.json"), str(tmp_path / "o2.bin")], capture_output=True, text=True)
    assert r.returncode == 6 and "rows*cols" in r.stderr


def _parse_only(tmp_path, text):
    (tmp_path / "t.json").write_text(text)
    r = subprocess.run([str(LIB / "cerebro_replay"), "--parse-only", str(tmp_path / "t.json"), str(tmp_path / "t.bin")], capture_output=True, text=True)
    if r.returncode != 0:
        return r.returncode, r.stderr
    raw = (tmp_path / "t.bin").read_bytes()
    Dg, n = struct.unpack_from("<IQ", raw, 0)
    return 0, np.frombuffer(raw, dtype=np.float64, count=n * Dg, offset=12 + 8 * n).reshape(n, Dg)


def test_state_json_number_conversion_edge_cases(tmp_path):
    """The data text is converted with from_chars and, for what it rejects, strtod -- together the reference's std::stod
    (RawFileIO.cpp:418-459).  Separators: real or escaped newlines/tabs, ", " (Eigen's coeff separator), and \\uXXXX."""
    vals = ["1.5", "-2.25e-3", "0.1", "123456789012345678", "4.9406564584124654e-324", "1e400", "-nan", "inf", "+3.5", "0x1p-2", "1E5", ".5"]
    want = np.array([float(v.replace("-nan", "nan")) if not v.startswith("0x") else float.fromhex(v) for v in vals])
    node = lambda data: json.dumps({"DataNodes": [{"stampNSec": 7, "wholeImageDescriptor": {"rows": len(vals), "cols": 1, "data": data}}]})
    for sep in ("\n", ", ", "\t \n", " "):
        rc, got = _parse_only(tmp_path, node(sep.join(vals)))
        assert rc == 0 and np.array_equal(got[0], want, equal_nan=True), (sep, got)
        assert np.signbit(got[0][8]) == False and got[0][5] == np.inf
    # the same text with every newline spelled as a \u escape takes the decode-first path
    doc = node("\n".join(vals)).replace("\\n", "\\u000a")
    assert "\\u000a" in doc
    rc, got = _parse_only(tmp_path, doc)
    assert rc == 0 and np.array_equal(got[0], want, equal_nan=True)
    # escaped quote / backslash pairs before the closing quote do not confuse the string scanner
    doc = json.dumps({"DataNodes": [{"stampNSec": 7, "note": "a\\\\", "q": "x\\\"y", "wholeImageDescriptor": {"rows": 2, "cols": 1, "data": "1\n2"}}]})
    rc, got = _parse_only(tmp_path, doc)
    assert rc == 0 and got.tolist() == [[1.0, 2.0]]
    for bad in ("1\n2\n3\n4", "1\nabc", "1\n2\n"):
        rc, err = _parse_only(tmp_path, json.dumps({"DataNodes": [{"stampNSec": 7, "wholeImageDescriptor": {"rows": 3, "cols": 1, "data": bad}}]}))
        assert rc == 6, (bad, err)
    # many descriptors -> several conversion threads; every row lands in its own slot
    N, D = 300, 8
    vals2 = np.arange(N * D, dtype=np.float64).reshape(N, D) / 7
    doc = json.dumps({"DataNodes": [{"stampNSec": i, "wholeImageDescriptor": {"rows": D, "cols": 1, "data": "\n".join(repr(float(x)) for x in vals2[i])}}
                                    for i in range(N)]})
    rc, got = _parse_only(tmp_path, doc)
    assert rc == 0 and np.array_equal(got, vals2)


def test_state_json_untrusted_input(tmp_path):
    """The loader does not trust the file: nodes come back in time order whatever the order of the array (the reference
    iterates the time-sorted data_map, Cerebro.cpp:145-149), absurd rows*cols and runaway nesting are errors, not crashes."""
    D = 4
    rows = {30: [3.0] * D, 10: [1.0] * D, 20: [2.0] * D, 15: [1.5] * D}
    doc = json.dumps({"DataNodes": [{"stampNSec": s, "wholeImageDescriptor": {"rows": D, "cols": 1, "data": "\n".join(map(repr, v))}}
                                    for s, v in rows.items()]})
    (tmp_path / "t.json").write_text(doc)
    r = subprocess.run([str(LIB / "cerebro_replay"), "--parse-only", str(tmp_path / "t.json"), str(tmp_path / "t.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "t.bin").read_bytes()
    Dg, n = struct.unpack_from("<IQ", raw, 0)
    assert list(np.frombuffer(raw, dtype=np.uint64, count=n, offset=12)) == [10, 15, 20, 30]
    assert np.frombuffer(raw, dtype=np.float64, count=n * Dg, offset=12 + 8 * n).reshape(n, Dg)[:, 0].tolist() == [1.0, 1.5, 2.0, 3.0]
    for rows_, cols_ in ((1 << 31, 2), (3_000_000_000, 3_000_000_000), (10241, 1), (4096, 4096)):
        rc, err = _parse_only(tmp_path, json.dumps({"DataNodes": [{"stampNSec": 7, "wholeImageDescriptor": {"rows": rows_, "cols": cols_, "data": "1"}}]}))
        assert rc == 6 and "out of range" in err, (rows_, cols_, err)
    rc, err = _parse_only(tmp_path, '{"junk": ' + "[" * 100000 + "]" * 100000 + ', "DataNodes": []}')
    assert rc == 6 and "nesting" in err
    rc, got = _parse_only(tmp_path, '{"junk": ' + "[" * 60 + "]" * 60 + ', "DataNodes": []}')
    assert rc == 0


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,devices,gaps", [(1024, 700, None, False), (256, 4600, "0,0,0", False), (512, 900, None, True)])
def test_cold_start_from_state_json_matches_oracle(tmp_path, D, N, devices, gaps):
    """--devices + a checkpoint longer than the replicated ring (CHIP_RING_ROWS = 4096): after the cold start the schedule starts at
    l = 56, 4500 rows behind the append head -- those ticks fetch their query rows from the sub-contexts that own them."""
    plants, loops, ties = scenarios.loop_plants(N, 4, seed=8)
    db = scenarios.build_db(19, N, D, plants)
    node_of_row = list(range(N))
    if gaps:
        # a checkpoint holds a node for EVERY camera frame; only keyframes carry a descriptor (here: two descriptor-less frames after
        # every third keyframe).  global_a / global_b of the dump are then the NODE index (= the reference's data_map index,
        # Cerebro.cpp:1142-1143), not the DB row.
        node_of_row, k = [], 0
        for i in range(N):
            node_of_row.append(k)
            k += 3 if i % 3 == 2 else 1
        n_nodes = k
        full = np.zeros((n_nodes, D), dtype=db.dtype)
        has = [False] * n_nodes
        for i, nd in enumerate(node_of_row):
            full[nd] = db[i]
            has[nd] = True
        node_stamps = [1403636579_000000000 + i * 50_000_000 for i in range(n_nodes)]
        write_state_json(tmp_path / "state.json", full, node_stamps, has)
        stamps = [node_stamps[nd] for nd in node_of_row]
    else:
        stamps = [1403636579_000000000 + i * 50_000_000 for i in range(N)]
        write_state_json(tmp_path / "state.json", db, stamps)
    r = subprocess.run([str(LIB / "cerebro_replay")] + (["--devices", devices] if devices else []) +
                       ["--state", str(tmp_path / "state.json"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads((tmp_path / "o.json").read_text())
    orc = oracle_lib.LoopOracle(db)                                     # the float32 descriptors the checkpoint was made from
    want = []
    for l in range(56, N + 1, 3):
        o = orc.tick(l)
        if o["found"]:
            want.append((node_of_row[o["idx_curr"]], node_of_row[o["idx_prev"]], o["score"]))
    assert [(g["global_a"], g["global_b"], g["score"]) for g in got] == want and len(want) >= len(loops)
    row0 = node_of_row.index(want[0][0])
    assert got[0]["time_sec_a"] == stamps[row0] // 10**9 and got[0]["time_nsec_a"] == stamps[row0] % 10**9


# ------------------------------------------------------------------ N4: top-k candidate policies over the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("policy,step,devices", [("naive", 3, None), ("clique", 1, None), ("clique", 3, None), ("clique", 5, None),
                                                 ("naive", 3, "0,0"), ("clique", 3, "0,0,0,0")])
def test_policy_replay_matches_oracle(tmp_path, policy, step, devices):
    import test_oracle_policies as pol
    db, plants = pol.policy_db(seed=9)
    N = db.shape[0]
    stamps = [(1403636579 + i // 20, (i % 20) * 50_000_000) for i in range(N)]
    ticks = list(range(step, N + 1, step))
    write_stream(tmp_path / "s.bin", db, stamps, ticks)
    r = subprocess.run([str(LIB / "cerebro_replay")] + (["--devices", devices] if devices else []) +
                       ["--policy", policy, str(tmp_path / "s.bin"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads((tmp_path / "o.json").read_text())
    orc = oracle_lib.NaivePolicyOracle(db) if policy == "naive" else oracle_lib.CliquePolicyOracle(db, oracle_lib.AnsiRand())
    want = [x for l in ticks for x in orc.tick(l)]
    assert len(want) >= 3
    assert [(g["global_a"], g["global_b"], g["score"]) for g in got] == want      # indices and float scores bit-exact


# ------------------------------------------------------------------ the INTEGRATION.md call sequence as a plain C-ABI program
def test_example_program_is_built_against_the_c_abi_only():
    exe = LIB / "minimal_loop_detector"
    assert exe.exists()
    out = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libcerebro_hip.so" in out and "libcerebro_host" not in out and "not found" not in out


@pytest.mark.gpu
def test_example_program_detects_the_revisit_and_recovers_the_pose():
    r = subprocess.run([str(LIB / "minimal_loop_detector")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    single = [l for l in r.stdout.splitlines() if l.startswith("loop candidate")]
    # the same program over chip_create_multi: a one-device group (RCCL communicator) and four shards on one device
    for devs in ("0", "0,0,0,0"):
        rm = subprocess.run([str(LIB / "minimal_loop_detector"), devs], capture_output=True, text=True, timeout=300)
        assert rm.returncode == 0, rm.stdout + rm.stderr
        assert [l for l in rm.stdout.splitlines() if l.startswith("loop candidate")] == single
    assert r.returncode == 0, r.stdout + r.stderr
    assert "loop candidate" in r.stdout and "PnP:" in r.stdout
