# round 3, call S: soak -- 3000 headline ticks (results checked against the planted schedule by bench.py), 5000 PnP calls with changing seeds
# compared against themselves for determinism (same seed twice -> same bits), and the 10k config for 20000 ticks
mkdir -p gpurun_out
timeout 600 python bench.py --steps 3000 --warmup 10 --cpu-budget 0 --no-pnp --no-batch --no-sizes 2>&1 | tail -1 | cut -c1-330
timeout 600 python scripts/gpu_short_scan.py --rows 10000 --ticks 20000 --set half 2>&1 | head -2 | cut -c1-300
timeout 600 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from cerebro_amd import capi
from cerebro_amd.synth import make_scene
X, uv, T, inl = make_scene(N=512, outlier_frac=0.3, noise_px=0.5, seed=4242)
with capi.Chip(64) as chip:
    p = capi.default_ransac_params(); p.n_hypotheses = 0
    ref = {}
    t0 = time.perf_counter()
    for i in range(5000):
        p.seed = 1000 + (i % 250)
        r = chip.pnp_ransac(X, uv, p)
        key = p.seed
        sig = (r["T"].tobytes(), r["mask"].tobytes(), r["confidence"], tuple(sorted(r["summary"].items())))
        if key in ref: assert ref[key] == sig, f"call {i}: seed {key} gave different bits"
        else: ref[key] = sig
    dt = time.perf_counter() - t0
print(f"PnP soak: 5000 reference-mode calls, 250 seeds x 20 repeats bit-identical, {dt/5000*1e6:.0f} us per call")
PY
