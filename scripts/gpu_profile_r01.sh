# Round-1 evidence run: default bench + rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/*
(timeout 900 python bench.py > gpurun_out/prof/bench_default.log 2>&1; echo exit=$? >> gpurun_out/prof/bench_default.log)
tail -2 gpurun_out/prof/bench_default.log | cut -c1-600
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r01 -- python bench.py --cpu-budget 0 > gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/pmc_fetch -o r01 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/pmc_write -o r01 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp > gpurun_out/prof/pmc_write.log 2>&1
grep '^{' gpurun_out/prof/trace.log | cut -c1-300
ls -la gpurun_out/prof/*/
