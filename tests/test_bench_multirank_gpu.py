"""bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run, one rank per "GPU"), on the
1-GPU test box: BENCH_DIST_BACKEND=gloo lets the ranks share device 0.  Checks the multi-rank plumbing end to end -- rank 0
prints exactly one JSON line with the contract's keys, the planted revisits are found through the sharded path."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("n,extra", [(2, []), (3, ["--replicated"])])
def test_bench_under_torchrun(n, extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
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
    assert j["scaling"] == ("weak" if extra else "strong")
    assert "cpu_baseline" not in j and "pnp" not in j          # rank 0 at N = 1 only
