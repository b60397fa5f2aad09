#!/usr/bin/env python3
"""Shader-clock split of the QR iteration of pnp_eig_score (CHIP_PNP_STAMPS=1 -> pnp_eig_score<true>): per wave, cycles spent in
the sweep overhead, the reflector (sqrt + division chain), the row modification and the column modification (+ forwarding), with
the numbers of sweeps and double-shift steps.  Config 3 scene: 512 correspondences, H hypotheses (default 1000)."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
os.environ["CHIP_PNP_STAMPS"] = "1"
from cerebro_amd import capi  # noqa: E402
from cerebro_amd.synth import make_scene  # noqa: E402

X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = H; p.seed = 4242
    for _ in range(3):
        chip.pnp_ransac(X, uv, p)
    t0 = time.perf_counter()
    for _ in range(10):
        chip.pnp_ransac(X, uv, p)
    call_us = (time.perf_counter() - t0) / 10 * 1e6
    fn = chip.lib.chip_debug_pnp_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    buf = np.zeros((H, 8), dtype=np.uint64)
    assert fn(chip.h, buf.ctypes.data, H) == 0
t = buf.astype(np.float64)
live = t[:, 7] > 0
t = t[live]
steps, sweeps = t[:, 4], t[:, 5]
print(f"H={H}: {live.sum()} waves ran the QR; whole call {call_us:.0f} us WITH stamps (the stamps cost ~10 %)")
print(f"double-shift steps per matrix: mean {steps.mean():.0f}  p95 {np.percentile(steps, 95):.0f}  max {steps.max():.0f};  sweeps: mean {sweeps.mean():.1f}  max {sweeps.max():.0f}")
tot = t[:, 0] + t[:, 1] + t[:, 2] + t[:, 3]
for name, col in (("sweep overhead (deflation test, shifts, m search)", 0), ("reflector: |p|+|q|+|r| .. sqrt .. division .. readlanes", 1),
                  ("row modification (LDS read, 5 dependent ops, write)", 2), ("column modification + forwarding", 3)):
    per = t[:, col] / np.where(col == 0, sweeps, steps)
    print(f"  {name:58s}: {100 * t[:, col].sum() / tot.sum():5.1f} % of the QR cycles, {per.mean():7.1f} cycles per {'sweep' if col == 0 else 'step'}")
slow = int(np.argmax(t[:, 7]))
print(f"kernel = slowest wave: {t[slow, 7]:.0f} cycles ({steps[slow]:.0f} steps, {sweeps[slow]:.0f} sweeps): Hessenberg {t[slow, 6]:.0f}, QR {tot[slow]:.0f} "
      f"= reflector {t[slow, 1]:.0f} + row {t[slow, 2]:.0f} + column {t[slow, 3]:.0f} + sweep overhead {t[slow, 0]:.0f}")
print(f"mean wave: {t[:, 7].mean():.0f} cycles, Hessenberg {t[:, 6].mean():.0f}, QR {tot.mean():.0f}; cycles per step (all QR cycles / steps): {(tot / steps).mean():.0f}")

# ---- pnp_build_solve: s_memtime at the phase boundaries (thread 0 of every workgroup)
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = H; p.seed = 4242
    for _ in range(3):
        chip.pnp_ransac(X, uv, p)
    fn = chip.lib.chip_debug_pnp_solve_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    buf = np.zeros((H, 24), dtype=np.uint64)
    assert fn(chip.h, buf.ctypes.data, H) == 0
t = buf.astype(np.float64)
t = t[t[:, 6] > 0]
names = ["sampler, bearings, H, W, T", "cost matrix M9 -> G -> quartic -> cubics", "Macaulay fill (registers)", "blocked LU (24 panels)",
         "back-substitution (27 unknowns x 27 RHS)", "S = A - B X, stores"]
tot = t[:, 6] - t[:, 0]
print(f"pnp_build_solve, {len(t)} workgroups: {tot.mean():.0f} cycles per workgroup (max {tot.max():.0f})")
for i, nme in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nme:44s}: {d.mean():9.0f} cycles ({100 * d.mean() / tot.mean():5.1f} %)")

print("  LU per phase, cycles per workgroup (sum over the 24 panels):")
print(f"    matrix wave 0: P3 (publish pivot rows) {t[:, 8].mean():8.0f}  barrier {t[:, 9].mean():8.0f}  P4 (trailing update) {t[:, 10].mean():8.0f}  barrier {t[:, 11].mean():8.0f}")
print(f"    factor wave 6: (idle in P3)            {t[:, 12].mean():8.0f}  barrier {t[:, 13].mean():8.0f}  F  (update + factor)   {t[:, 14].mean():8.0f}  barrier {t[:, 15].mean():8.0f}")

fn_ = ["update of the panel by the previous one (+ waits)", "candidates + wave max", "compare / ballots / logical position", "pivot-row + multiplier readlanes",
       "two IEEE divisions", "panel update + logical positions", "publication (LDS writes)", "-"]
if t[:, 16:23].sum() > 0:      # only in a -DCHIP_PNP_FSTAMPS build
  print("  factor wave, cycles per workgroup (24 panels, 93 columns):")
  for i in range(7):
    print(f"    {fn_[i]:52s}: {t[:, 16 + i].mean():9.0f}  ({t[:, 16 + i].mean() / (24 if i in (0, 6) else 93):7.0f} per {'panel' if i in (0, 6) else 'column'})")

hw = buf[buf[:, 6] > 0][:, 23]
from collections import Counter
cnt = Counter(tuple(int(h >> (4 * w)) & 3 for w in range(7)) for h in hw)
print("  SIMD of waves 0..6 (HW_ID), most common placements:", cnt.most_common(4))
