set -x
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_exit=$? >> gpurun_out/pytest_gpu.log)
tail -8 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log)
tail -3 gpurun_out/bench_default.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r01 -- python bench.py --steps 30 --warmup 5 --cpu-budget 0 > gpurun_out/prof/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/pmc_fetch -o r01 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/pmc_write -o r01 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 > gpurun_out/prof/pmc_write.log 2>&1
find gpurun_out/prof -type f | head -50
du -sh gpurun_out/prof
