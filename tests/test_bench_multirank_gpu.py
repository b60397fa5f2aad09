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
                                          (2, [], {"BENCH_HANG_COMM_INIT": "1", "BENCH_COMM_INIT_TIMEOUT": "2"}),
                                          # double fault: the in-library attach FAILS, and the torch.distributed nccl group the run falls back to
                                          # cannot run its first collective either (here: RCCL refuses two ranks on one device) -- that collective
                                          # is the warmup, which runs under a deadline; the run ends on the gloo exchange, with its JSON line
                                          (2, [], {"BENCH_FAIL_COMM_INIT": "1", "BENCH_WARMUP_TIMEOUT": "30"})])
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
    if "BENCH_HANG_COMM_INIT" in hook:
        assert "hung" in j["config"]["exchange"]
    if "BENCH_FAIL_COMM_INIT" in hook:
        assert "gloo" in j["config"]["exchange"] and "torch.distributed nccl" in j["config"]["exchange"] and j["config"]["exchange_runtime_fallback"]
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
        assert "ms_per_step_median" not in leg                 # a collection cadence, not a rate
        if k != "200k":     # the size legs carry BOTH accountings under names that say what they are
            r = leg["roofline"]
            alg = r["algorithmic_bytes_per_launch"]
            assert abs(r["achieved_step"] - alg / (leg["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * r["achieved_step"]
            assert abs(r["achieved_kernel"] - alg / (r["isolated_kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved_kernel"]
            assert r["achieved"] == r["achieved_kernel"] and r["frac"] == r["frac_kernel"]
            assert 0 < r["frac_kernel"] < 1 and 0 < r["frac_step"] < 1        # nothing implies more than the 8 TB/s peak
            assert leg["sync_tick_us"] >= leg["sync_tick_us_min"] > 0
            assert leg["sync_tick_us_min"] * 1e-3 >= 0.9 * r["isolated_kernel_ms"] * 0.5   # a synchronous tick contains its kernel
    assert j["config"]["exchange"] == "none" and j["details"]["rccl_ranks"] == 0 and j["details"]["exchange_fallback"] is False
    assert len(j["config"]) <= 24 and list(j["config"])[0] == "workload"
    assert list(j["config"])[1:4] == ["size_10k_roofline_frac_kernel", "size_29k_roofline_frac_kernel", "size_100k_roofline_frac_kernel"]   # --no-pnp: the size scalars lead
    assert j["roofline"]["size_29k_roofline_frac_kernel"] == j["sizes"]["29k"]["roofline"]["frac_kernel"]
    assert j["sizes"]["10k"]["roofline"]["cache_resident"] and not j["sizes"]["100k"]["roofline"]["cache_resident"]
    assert j["roofline"]["traffic"] is None and "ms_per_step_median" not in j
    j64 = _run_single(["--rows", "60000", "--storage", "f64", "--no-sizes"])
    assert "fp64 rows" in j64["config"]["storage"] and j64["roofline"]["algorithmic_bytes_per_launch"] == 4.0 * 4096 * 60000


def _with_env(env, args):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _run_single(args)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_bench_group_mode_survives_a_hung_or_failed_rccl_bootstrap():
    """`python bench.py --gpus N` without torchrun = ONE process over chip_create_multi -> ncclCommInitAll.  A bootstrap that hangs
    or fails must cost the run its RCCL exchange, never its JSON line.  --force-group runs that layout on the
    1-GPU box (devices = [0]: RCCL transport, one rank)."""
    base = ["--force-group", "--rows", "60000"]
    j = _with_env({"CHIP_TEST_COMM_INIT": "fail"}, base)    # healthy, and the PRODUCT build does not even read the fault-injection variable
    assert j["details"]["test_hooks"] == 0
    assert "in-library RCCL (ncclCommInitAll" in j["config"]["exchange"] and j["config"]["rccl_ranks"] == 1
    assert j["config"]["exchange_fallback"] is False and j["config"]["comm_init_abandoned"] is False
    # the library's own deadline: ncclCommInitAll never returns -> abandoned on its helper thread, device-copy exchange
    from conftest import HOOKS_ENV          # fault injection exists in the TEST build of the library only (cerebro_amd/lib/hooks/)
    j = _with_env({"CHIP_TEST_COMM_INIT": "hang", "BENCH_COMM_INIT_TIMEOUT": "2", **HOOKS_ENV}, base)
    assert j["details"]["test_hooks"] == 1
    assert "FALLBACK" in j["config"]["exchange"] and "hung" in j["config"]["exchange"]
    assert j["config"]["rccl_ranks"] == 0 and j["config"]["exchange_fallback"] is True and j["config"]["comm_init_abandoned"] is True and j["value"] > 0
    # ncclCommInitAll fails outright
    j = _with_env({"CHIP_TEST_COMM_INIT": "fail", **HOOKS_ENV}, base)
    assert "FALLBACK" in j["config"]["exchange"] and j["config"]["rccl_ranks"] == 0 and j["config"]["exchange_fallback"] is True and j["value"] > 0
    # something else inside the create hangs: bench.py's outer deadline rebuilds the group on the copy exchange
    j = _with_env({"BENCH_HANG_GROUP_CREATE": "1", "BENCH_GROUP_CREATE_TIMEOUT": "2"}, base)
    assert "FALLBACK" in j["config"]["exchange"] and "create hung" in j["config"]["exchange"] and j["config"]["exchange_fallback"] is True and j["value"] > 0


def test_bench_survives_a_first_collective_that_never_completes():
    """The bootstrap succeeds but the first ncclAllGather (the warmup) never comes back: the warmup runs under a deadline and the run
    is rebuilt on an exchange that needs no RCCL -- group mode: device copies; one process per GPU: host-driven over gloo."""
    j = _with_env({"BENCH_HANG_WARMUP": "1", "BENCH_WARMUP_TIMEOUT": "3"}, ["--force-group", "--rows", "60000"])
    assert "FALLBACK: the warmup over RCCL did not finish" in j["config"]["exchange"] and j["config"]["exchange_fallback"] is True
    assert j["config"]["exchange_runtime_fallback"] and j["value"] > 0
    j = _with_env({"BENCH_HANG_WARMUP": "1", "BENCH_WARMUP_TIMEOUT": "3"}, ["--force-sharded", "--rows", "60000"])
    assert "host-driven fallback" in j["config"]["exchange"] and "gloo" in j["config"]["exchange"] and j["config"]["exchange_fallback"] is True and j["value"] > 0
