"""bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run, one rank per "GPU"), on the
1-GPU test box: the ranks share device 0, so the exchange is host-driven over gloo (--host-exchange; RCCL refuses two ranks on
one device -- the in-library RCCL exchange the driver's 8-GPU run uses is covered at world size 1 by --force-sharded below and
by tests/test_multi_gpu.py).  Checks the multi-rank plumbing end to end -- rank 0 prints exactly one JSON line with the
contract's keys, the planted revisits are found through the sharded path."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("n,extra,hook", [(2, ["--host-exchange"], {}), (3, ["--replicated"], {}),
                                          # the in-library attach "hangs" on every rank: deadline -> exchange over gloo, exit code 0
                                          (2, [], {"BENCH_HANG_COMM_INIT": "1", "BENCH_COMM_INIT_TIMEOUT": "2"})])
def test_bench_under_torchrun(n, extra, hook):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", **hook)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "24", "--warmup", "4", "--rows", "60000"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == n and j["steps"] == 24 and j["value"] > 0
    assert j["scaling"] == ("weak" if "--replicated" in extra else "strong")
    if hook:
        assert "hung" in j["config"]["exchange"]
    assert "cpu_baseline" not in j and "pnp" not in j          # rank 0 at N = 1 only


def _run_single(args):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "24", "--warmup", "4", "--cpu-budget", "0", "--no-pnp", "--no-batch"] + args,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_one_process_group_same_device():
    """--gpus 4 without torchrun: ONE process over chip_create_multi; --same-device puts the four sub-contexts on device 0."""
    j = _run_single(["--gpus", "4", "--same-device", "--rows", "60000"])
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["value"] > 0 and j["config"]["process_layout"] == "one process"
    assert "device copies" in j["config"]["exchange"] and "sizes" not in j


def test_bench_force_sharded_uses_in_library_rccl():
    j = _run_single(["--force-sharded", "--rows", "60000"])
    assert j["n_gpus"] == 1 and "in-library RCCL" in j["config"]["exchange"] and j["value"] > 0
    # the agreed fallback when attaching the communicator fails on some rank: host-driven exchange, same answers
    os.environ["BENCH_FAIL_COMM_INIT"] = "1"
    try:
        j = _run_single(["--force-sharded", "--rows", "60000"])
    finally:
        del os.environ["BENCH_FAIL_COMM_INIT"]
    assert "host-driven fallback" in j["config"]["exchange"] and j["value"] > 0
    # ... and when the attach never returns (an RCCL bootstrap that hangs): deadline, exchange over the control plane, clean exit
    os.environ["BENCH_HANG_COMM_INIT"] = "1"
    os.environ["BENCH_COMM_INIT_TIMEOUT"] = "2"
    try:
        j = _run_single(["--force-sharded", "--rows", "60000"])
    finally:
        del os.environ["BENCH_HANG_COMM_INIT"], os.environ["BENCH_COMM_INIT_TIMEOUT"]
    assert "hung" in j["config"]["exchange"] and "gloo" in j["config"]["exchange"] and j["value"] > 0


def test_bench_sizes_key_and_f64_storage():
    j = _run_single(["--rows", "200000"])
    assert set(j["sizes"]) == {"10k", "29k", "100k", "200k"}
    for k, leg in j["sizes"].items():
        assert leg["value"] > 0 and leg["roofline"]["achieved"] > 0 and leg["roofline"]["algorithmic_bytes_per_launch"] == 4.0 * 4096 * leg["db_rows"]
        if k != "200k":     # the size legs are priced from their step time and say so
            r = leg["roofline"]
            assert r["kernel_overlap"] is True and abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (leg["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
            assert r["isolated_kernel_ms"] > 0
    assert j["config"]["rccl_ranks"] == 0 and j["config"]["exchange_fallback"] is False
    assert j["sizes"]["10k"]["roofline"]["cache_resident"] and not j["sizes"]["100k"]["roofline"]["cache_resident"]
    assert j["roofline"]["traffic"] is None and j["ms_per_step_median"] > 0
    j64 = _run_single(["--rows", "60000", "--storage", "f64", "--no-sizes"])
    assert "fp64 rows" in j64["config"]["storage"] and j64["roofline"]["algorithmic_bytes_per_launch"] == 4.0 * 4096 * 60000
