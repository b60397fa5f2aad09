# the two bench lines of the final tree (the driver's command, timed; the default command)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err ) 2> $O/bench_driver.time; tail -3 $O/bench_driver.time; tail -1 $O/bench_driver.log | cut -c1-200
( time python bench.py > $O/bench_default.log 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; tail -1 $O/bench_default.log | cut -c1-200
