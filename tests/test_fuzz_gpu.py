"""Randomised GPU-vs-oracle sweeps (the generators live in scripts/gpu_*_fuzz.py, which run the same thing at larger counts):
odd PnP/ICP scenes (planar, duplicated points, extreme scale, heavy outliers, minimal N) and odd scan shapes (random D, prefix,
nq, topk, exact ties, duplicated rows, extreme magnitudes, the MFMA mode).  Everything bit for bit."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))

pytestmark = pytest.mark.gpu


def test_pnp_icp_fuzz():
    import gpu_pnp_fuzz
    bad, n_models = gpu_pnp_fuzz.run(64)
    assert not bad, bad[:3]
    assert n_models > 500          # the sweep does exercise the solver, not only rejections


def test_scan_fuzz():
    import gpu_scan_fuzz
    bad = gpu_scan_fuzz.run(60)
    assert not bad, bad[:3]


def test_tick_schedule_fuzz():
    import gpu_scan_fuzz
    bad = gpu_scan_fuzz.run_ticks(6)
    assert not bad, bad[:3]


def test_f64_and_group_fuzz():
    import gpu_scan_fuzz
    bad = gpu_scan_fuzz.run_f64_and_groups(40)
    assert not bad, bad[:3]
