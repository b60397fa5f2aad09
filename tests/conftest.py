import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no built artefacts (they are git-ignored): build them once, exactly as __graft_entry__.build() does.
    needed = [ROOT / "cerebro_amd" / "lib" / n for n in ("libcerebro_hip.so", "libcerebro_host.so", "cerebro_replay", "minimal_loop_detector")]
    needed.append(ROOT / "oracle" / "_build" / "liboracle.so")
    if not all(p.exists() for p in needed):
        import subprocess
        r = subprocess.run(["make", "-j4", "all"], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            raise pytest.UsageError("make all failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def chip_lib():
    from cerebro_amd import capi
    return capi.load_library()
