# round 3, call I: evidence run -- default bench, rocprofv3 kernel-trace stats of the same command, separate PMC passes (FETCH_SIZE / WRITE_SIZE)
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/*
timeout 300 python scripts/gpu_group_append.py 100000 > gpurun_out/group_append.txt 2>&1; grep wire gpurun_out/group_append.txt
(timeout 900 python bench.py > gpurun_out/prof/bench_default.log 2>&1; echo exit=$? >> gpurun_out/prof/bench_default.log)
tail -2 gpurun_out/prof/bench_default.log | cut -c1-400
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r03 -- python bench.py --cpu-budget 0 > gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/pmc_fetch -o r03 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/pmc_write -o r03 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/prof/pmc_write.log 2>&1
grep '^{' gpurun_out/prof/trace.log | cut -c1-200
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_i.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_i.log); tail -4 gpurun_out/pytest_i.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl"
