import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


HIP_ARTEFACTS = [ROOT / "cerebro_amd" / "lib" / n for n in ("libcerebro_hip.so", "libcerebro_host.so", "cerebro_replay", "minimal_loop_detector", "sync_tick_latency",
                                                          "norows/libcerebro_hip.so", "hooks/libcerebro_hip.so")] + [ROOT / "tests" / "fakerccl" / "_build" / "libfakerccl.so"]
_hip_build_error = None


def pytest_configure(config):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, as __graft_entry__.build() does.
    The CPU oracle is needed by every test and must build; the HIP side needs hipcc -- without it (or if it fails) only
    the tests that load libcerebro_hip.so / run the host binaries are skipped, the oracle / gloo / parser tests still run."""
    global _hip_build_error
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_hip_build: loads libcerebro_hip.so or runs a host binary linked against it")
    import shutil
    import subprocess
    if not (ROOT / "oracle" / "_build" / "liboracle.so").exists():
        r = subprocess.run(["make", "oracle"], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            raise pytest.UsageError("make oracle failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    if not all(p.exists() for p in HIP_ARTEFACTS):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not (os.path.exists(hipcc) or shutil.which("hipcc")):
            _hip_build_error = "hipcc not found: libcerebro_hip.so cannot be built on this machine"
        else:
            r = subprocess.run(["make", "-j4", "lib", "host", "testlibs"], cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0:
                _hip_build_error = "make lib host testlibs failed:\n" + r.stdout[-1500:] + r.stderr[-1500:]


def pytest_collection_modifyitems(config, items):
    if _hip_build_error is None:
        return
    skip = pytest.mark.skip(reason=_hip_build_error)
    for it in items:
        if "gpu" in it.keywords or "needs_hip_build" in it.keywords or "chip_lib" in getattr(it, "fixturenames", ()):
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture
def hooks_lib():
    """Chip objects created by this test run on the TEST build of the library (cerebro_amd/lib/hooks/, -DCHIP_TEST_HOOKS): the only
    build that contains the fault-injection hooks.  Every other test -- and every product path -- runs the product build, which reads
    none of the CHIP_TEST_* variables."""
    from cerebro_amd import capi
    with capi.use_hooks_library() as lib:
        yield lib


HOOKS_ENV = {"CHIP_LIB": str(ROOT / "cerebro_amd" / "lib" / "hooks" / "libcerebro_hip.so"), "CHIP_ALLOW_LIB_OVERRIDE": "1"}   # for subprocesses


@pytest.fixture(scope="session")
def chip_lib():
    from cerebro_amd import capi
    return capi.load_library()
