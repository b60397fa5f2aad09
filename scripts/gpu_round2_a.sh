# Round-2 evidence run A: full GPU suite, default bench, kernel-trace stats, multi-GPU host-cost probe
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; echo exit=$? >> gpurun_out/r02/pytest_gpu.log)
tail -5 gpurun_out/r02/pytest_gpu.log
(timeout 900 python bench.py > gpurun_out/r02/bench_default.log 2>&1; echo exit=$? >> gpurun_out/r02/bench_default.log)
tail -2 gpurun_out/r02/bench_default.log | cut -c1-1500
(timeout 600 python scripts/gpu_group_cost.py > gpurun_out/r02/group_cost.json 2>gpurun_out/r02/group_cost.err; echo exit=$?)
cat gpurun_out/r02/group_cost.json | head -60
rm -rf gpurun_out/r02/trace
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r02/trace -o r02 -- python bench.py --cpu-budget 0 > gpurun_out/r02/trace.log 2>&1
grep '^{' gpurun_out/r02/trace.log | cut -c1-300
ls gpurun_out/r02/trace | head
