"""The C++ host mirror (cerebro_amd/host) and the replay harness: build/link checks on CPU, end-to-end replay on GPU."""
import json
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import scenarios

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
def test_replay_matches_oracle(tmp_path):
    D, N = 512, 1300
    plants, loops, ties = scenarios.loop_plants(N, 6, seed=21)
    db = scenarios.build_db(77, N, D, plants)
    stamps = [(1403636579 + i // 20, (i % 20) * 50_000_000) for i in range(N)]     # 20 Hz keyframes
    ticks = scenarios.default_schedule(N)
    write_stream(tmp_path / "s.bin", db, stamps, ticks)
    r = subprocess.run([str(LIB / "cerebro_replay"), str(tmp_path / "s.bin"), str(tmp_path / "o.json")], capture_output=True, text=True)
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
