# round 3, call Q: evidence run on the final tree -- full GPU suite, default bench, rocprofv3 kernel-trace stats of the same command,
# separate PMC passes (FETCH_SIZE / WRITE_SIZE), PnP stamps + rates, chain-latency microbenchmark, multi-rank benches on one device
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/*
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_q.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_q.log); tail -4 gpurun_out/pytest_q.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
(timeout 900 python bench.py > gpurun_out/prof/bench_default.log 2>&1; echo exit=$? >> gpurun_out/prof/bench_default.log)
tail -2 gpurun_out/prof/bench_default.log | cut -c1-600
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r03 -- python bench.py --cpu-budget 0 > gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/pmc_fetch -o r03 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/pmc_write -o r03 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/prof/pmc_write.log 2>&1
grep '^{' gpurun_out/prof/trace.log | cut -c1-200
timeout 300 python scripts/gpu_pnp_stamps.py 50 > gpurun_out/pnp_stamps_q.txt 2>&1
for i in 1 2 3; do timeout 300 python scripts/gpu_pnp_rates.py 2>&1 | tail -1; done | tee gpurun_out/pnp_rates_q.txt
timeout 60 scripts/ubench/chain_latency > gpurun_out/chain_latency.txt 2>&1
(timeout 600 python bench.py --gpus 8 --same-device --no-pnp --no-batch --no-sizes --cpu-budget 0 2>&1 | tail -1 | cut -c1-700) | tee gpurun_out/bench_group8_same_device.log
