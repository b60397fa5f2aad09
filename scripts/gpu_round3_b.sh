# round 3, call B: where a 10k-row scan launch spends its time (per-wave stamps), the read ceiling of a cache-resident 164 MB buffer,
# host cost of a tick, the new config-4 tests, and a default bench run
mkdir -p gpurun_out
timeout 300 python scripts/gpu_scan_stamps.py 10000 > gpurun_out/scan_stamps_10k.txt 2>&1; tail -40 gpurun_out/scan_stamps_10k.txt
timeout 300 scripts/probes/hbm_read_probe.bin 0.1526 > gpurun_out/read_probe_164MB.txt 2>&1; grep -E "BEST|shape 1 U=8 nt|shape 1 U=8  " gpurun_out/read_probe_164MB.txt | head -20
timeout 120 python scripts/gpu_tick_host_times.py 10000 16 > gpurun_out/tick_host_10k.txt 2>&1; tail -2 gpurun_out/tick_host_10k.txt
CHIP_TICK_SAME_STREAM=0 timeout 120 python scripts/gpu_tick_host_times.py 10000 16 >> gpurun_out/tick_host_10k.txt 2>&1; tail -1 gpurun_out/tick_host_10k.txt
(timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_scan_gpu.py -m gpu -q -x > gpurun_out/pytest_b.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_b.log); tail -4 gpurun_out/pytest_b.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_b.json"))
print("value", d["value"], "roofline", d["roofline"]["frac"])
for k, v in d.get("sizes", {}).items(): print(k, v["value"], v["ms_per_step"], v["roofline"].get("frac"), v["roofline"].get("isolated_kernel_ms"))
print("pnp", d["pnp"]["value"], d["pnp"]["batch8_hypotheses_per_s"], "batch", d["batch"]["roofline"]["frac"])
print(d["config"]["rccl_ranks"], d["config"]["exchange_fallback"])
PY
