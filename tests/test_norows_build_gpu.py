"""The degraded build of the library (-DCHIP_NO_ROWS_FORM: what `make verify` falls back to when the code-object check of the
row-batched scan kernel fails on a different hipcc, scripts/verify_codeobj.sh) is a valid product: it says what it is
(chip_get_info().scan_forms) and passes the scan parity suite -- every scan through the one-row kernel, same bits."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
NOROWS = ROOT / "cerebro_amd" / "lib" / "norows" / "libcerebro_hip.so"


def test_full_build_has_both_forms():
    from cerebro_amd import capi
    with capi.Chip(256) as chip:
        assert chip.info()["scan_forms"] == capi.CHIP_SCAN_FORM_ONE_ROW | capi.CHIP_SCAN_FORM_ROWS
    assert capi.load_library().chip_build_scan_forms() == 3


def test_degraded_build_says_so_and_passes_the_scan_suite():
    assert NOROWS.exists(), "make testlibs builds cerebro_amd/lib/norows/libcerebro_hip.so"
    env = dict(os.environ, CHIP_LIB=str(NOROWS), CHIP_ALLOW_LIB_OVERRIDE="1")
    code = ("from cerebro_amd import capi\n"
            "with capi.Chip(4096) as c:\n"
            "    i = c.info(); assert i['scan_forms'] == capi.CHIP_SCAN_FORM_ONE_ROW, i\n"
            "    c.append_synthetic(3000, 1, []); r = c.loop_tick(3000); assert r.status == 2\n"
            "print('norows ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "norows ok" in r.stdout, r.stderr[-2000:]
    # the scan parity suite against that build (the 1M full-size case and the 200k-tick hand-off stress -- a property of the rows
    # form's fused tick, absent here -- stay with the full build)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_scan_gpu.py", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "not 1M_full_size and not handoff_stress"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
