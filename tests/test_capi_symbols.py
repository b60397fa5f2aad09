"""CPU-side checks of the drop-in boundary: libcerebro_hip.so loads, exports every symbol that
include/cerebro_hip.h declares, and fails loudly (status, not a CPU fallback) without a GPU."""
import ctypes as C
import re
from pathlib import Path

import pytest

from cerebro_amd import capi

pytestmark = pytest.mark.needs_hip_build   # uses libcerebro_hip.so / the host binaries (conftest skips these without hipcc)
ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "cerebro_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(chip_lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(chip_lib, s), f"{s} declared in cerebro_hip.h but not exported"
    assert sorted(capi.declared_symbols()) == syms      # the ctypes table binds exactly the header's surface


def test_abi_version_and_defaults(chip_lib):
    assert chip_lib.chip_abi_version() == 7
    p = capi.default_dot_params()
    assert (p.locality, p.lag, p.min_new, p.min_k) == (12, 50, 3, 5)        # Cerebro.cpp:912-914,962,1022
    assert p.thresh == 0.85000002384185791015625                           # (double)(float)0.85
    r = capi.default_ransac_params()
    assert (r.error_thresh, r.min_inlier_ratio, r.max_iterations, r.min_iterations, r.use_mle, r.sample_size) == \
        (0.03, 0.7, 50, 5, 1, 15)                                           # DlsPnpWithRansac.cpp:207-212, .h:45
    assert chip_lib.chip_strerror(0) == b"ok"
    assert b"float32" in chip_lib.chip_strerror(capi.CHIP_ERR_NOT_F32)


def test_struct_layouts_match_header():
    assert C.sizeof(capi.TickResult) == 4 + 4 + 8 + 8 + 8 + 24 + 24
    assert C.sizeof(capi.TopkEntry) == 16
    assert C.sizeof(capi.DotParams) == 24
    assert C.sizeof(capi.RansacParams) == 56
    assert C.sizeof(capi.RansacSummary) == 24


def test_invalid_arguments_are_status_codes(chip_lib):
    h = C.c_void_p()
    assert chip_lib.chip_create(None, 4096, 0, 0, 0, 1) == capi.CHIP_ERR_INVALID_ARG
    assert chip_lib.chip_create(C.byref(h), 0, 0, 0, 0, 1) == capi.CHIP_ERR_INVALID_ARG
    assert chip_lib.chip_create(C.byref(h), 4096, 0, 0, 2, 2) == capi.CHIP_ERR_INVALID_ARG
    assert chip_lib.chip_create(C.byref(h), 4098, 0, 0, 0, 1) == capi.CHIP_ERR_UNSUPPORTED   # D % 4 != 0


def test_no_gpu_fails_loudly(chip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert chip_lib.chip_create(C.byref(h), 4096, 0, 0, 0, 1) == capi.CHIP_ERR_NO_DEVICE
    with pytest.raises(capi.ChipError):
        capi.Chip(4096)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        capi.load_library(tmp_path / "libcerebro_hip.so")


HOOK_NAMES = [b"CHIP_TEST_COMM_INIT", b"CHIP_TEST_FAIL_SHARD", b"CHIP_TEST_BATCH_OOM", b"CHIP_TEST_RESIDENT_SKIP_MASTER", b"CHIP_TEST_RCCL_SAME_DEVICE", b"CHIP_PNP_BACKSUB", b"CHIP_PNP_DEBUG_STOP"]


def test_product_library_contains_no_test_hook(chip_lib):
    """`make lib` compiles the fault-injection hooks and the result-changing test knobs OUT -- the product .so does not
    even contain the names of their environment variables, so a stray variable cannot degrade a deployed node; the TEST build
    (`make testlibs` -> lib/hooks/, loaded by tests/ only) contains all of them and says so (chip_build_test_hooks / chip_info.test_hooks)."""
    assert chip_lib.chip_build_test_hooks() == 0
    blob = capi.PRODUCT_LIB_PATH.read_bytes()
    for name in HOOK_NAMES:
        assert name not in blob, name
    assert b"TEST HOOK ACTIVE" not in blob and b"TEST KNOB ACTIVE" not in blob
    hooks = capi.load_library(capi.HOOKS_LIB_PATH)
    assert hooks.chip_build_test_hooks() == 1 and hooks.chip_abi_version() == chip_lib.chip_abi_version()
    blob = capi.HOOKS_LIB_PATH.read_bytes()
    for name in HOOK_NAMES:
        assert name in blob, name
    # the other shipped variant (the degraded build) is a product build too
    assert capi.load_library(ROOT / "cerebro_amd" / "lib" / "norows" / "libcerebro_hip.so").chip_build_test_hooks() == 0


def test_chip_lib_override_needs_its_switch(tmp_path):
    """CHIP_LIB redirects the Python binding to another build only together with CHIP_ALLOW_LIB_OVERRIDE=1; alone it is ignored, loudly."""
    import os
    import subprocess
    import sys
    code = "from cerebro_amd import capi; print(capi.LIB_PATH)"
    other = str(capi.HOOKS_LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k not in ("CHIP_LIB", "CHIP_ALLOW_LIB_OVERRIDE")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, CHIP_LIB=other), capture_output=True, text=True)
    assert r.stdout.strip() == str(capi.PRODUCT_LIB_PATH) and "IGNORED" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, CHIP_LIB=other, CHIP_ALLOW_LIB_OVERRIDE="1"), capture_output=True, text=True)
    assert r.stdout.strip() == other and "IGNORED" not in r.stderr


def test_use_hooks_library_is_scoped():
    before = capi.load_library()
    with capi.use_hooks_library() as lib:
        assert capi.load_library() is lib and lib.chip_build_test_hooks() == 1
    assert capi.load_library() is before and before.chip_build_test_hooks() == 0
