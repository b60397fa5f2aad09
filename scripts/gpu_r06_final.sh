# final check of the tree: build entry, smoke, whole GPU suite, default bench (timed), the driver's bench command (timed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
( time python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
(timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log); grep -E "passed|failed|pytest_exit|^FAILED|^ERROR" $O/pytest_gpu.log | tail -8
( time python bench.py > $O/bench_default.log 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; tail -1 $O/bench_default.log | cut -c1-300
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err ) 2> $O/bench_driver.time; tail -3 $O/bench_driver.time
