# final tree, second evidence call: smoke(), the whole GPU suite with CHIP_TICK_RESIDENT=1, the DRIVER's bench command (timed), the PnP soak, a 3000-step bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06
( time python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
bash scripts/gpu_round6.sh suiteres
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err ) 2> $O/bench_driver.time; tail -3 $O/bench_driver.time; tail -1 $O/bench_driver.log | cut -c1-300
(timeout 900 python scripts/gpu_pnp_soak.py 250 20 2>&1 | grep -v amdgpu.ids; timeout 600 python bench.py --steps 3000 --cpu-budget 0 --no-pnp --no-batch --no-sizes --no-shapes 2>/dev/null | cut -c1-400) | tee $O/soak.txt
