mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_e.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_e.log); tail -12 gpurun_out/pytest_e.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_e.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"])
for k, v in d.get("sizes", {}).items(): print(k, v["value"], v["ms_per_step"], v["roofline"].get("frac"), v["roofline"].get("isolated_kernel_ms"))
print("pnp", d["pnp"]["value"], d["pnp"]["batch8_hypotheses_per_s"], "batch", d["batch"]["roofline"]["frac"])
PY
