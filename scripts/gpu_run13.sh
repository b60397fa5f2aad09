mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -6 gpurun_out/pytest_gpu.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import numpy as np
from cerebro_amd import capi
D=4096; n=40000
rows=(np.random.default_rng(0).standard_normal((n,D))/64).astype(np.float32)
r64=rows.astype(np.float64)
with capi.Chip(D, capacity_hint=4*n) as chip:
    chip.append_f32(rows[:100])
    t0=time.perf_counter(); chip.append_f32(rows); t1=time.perf_counter(); chip.append_f64(r64); t2=time.perf_counter()
    t3=time.perf_counter()
    for i in range(200): chip.append_f64(r64[i:i+1])
    t4=time.perf_counter()
    print(f"append_f32 bulk: {n*D*4/(t1-t0)/1e9:.1f} GB/s host->DB ({n/(t1-t0):.0f} rows/s); append_f64 bulk: {n*D*8/(t2-t1)/1e9:.1f} GB/s of f64 ({n/(t2-t1):.0f} rows/s); single-row append_f64: {(t4-t3)/200*1e6:.0f} us per keyframe")
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 1200 --warmup 5 --rows 100000 --cpu-budget 0 --no-pnp --no-batch 2>&1 | grep '^{' | cut -c1-330
timeout 300 python bench.py --steps 1200 --warmup 5 --rows 100000 --cpu-budget 0 --no-pnp --no-batch --force-sharded 2>&1 | grep '^{' | cut -c1-200
