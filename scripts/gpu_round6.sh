# round-6 GPU calls, one parameterised script:  bash scripts/gpu_round5.sh <step> [<step> ...]
#   suite      pytest -m gpu (whole parity suite)             -> gpurun_out/r06/pytest_gpu.log
#   suiteres   the same suite with CHIP_TICK_RESIDENT=1 in the environment (every synchronous tick of every test that qualifies goes through the
#              resident scan instance)                         -> gpurun_out/r06/pytest_gpu_resident.log
#   bench      default python bench.py                        -> gpurun_out/r06/bench_default.log
#   trace      rocprofv3 --kernel-trace --stats of the same   -> gpurun_out/r06/prof/trace (summarised by scripts/summarize_rocprof.py r06 gpurun_out/r06/prof)
#   scanpmc    FETCH_SIZE / WRITE_SIZE passes of the 1M tick  -> gpurun_out/r06/prof/pmc_*
#   sizespmc   scripts/gpu_scan_sizes_pmc.sh                  -> gpurun_out/sizes_pmc (summarised by scripts/summarize_sizes_pmc.py)
#   pnppmc     SQ counters of the batched PnP kernel pair     -> gpurun_out/r06/pnp_pmc.json
#   eigen      scripts/eigen_pin.py                           -> gpurun_out/r06/eigen_pin.json
#   icp        scripts/gpu_icp_perf.py                        -> gpurun_out/r06/icp_perf.txt
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
for step in "$@"; do
  case $step in
    suite) (timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log); grep -E "passed|failed|pytest_exit|Error" $O/pytest_gpu.log | tail -5 ;;
    suiteres) (CHIP_TICK_RESIDENT=1 timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_resident.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu_resident.log); grep -E "passed|failed|pytest_exit|Error" $O/pytest_gpu_resident.log | tail -5 ;;
    bench) (timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo exit=$? >> $O/bench_default.log); tail -2 $O/bench_default.log | cut -c1-600; tail -3 $O/bench_default.err ;;
    trace) rm -rf $O/prof/trace; timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof/trace -o r06 -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/trace.log 2>&1; grep '^{' $O/trace.log | cut -c1-200 ;;
    scanpmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/prof/pmc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/prof/pmc_$c -o r06 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > $O/pmc_$c.log 2>&1; done ;;
    sizespmc) bash scripts/gpu_scan_sizes_pmc.sh ;;
    pnppmc) bash scripts/gpu_pnp_pmc5.sh > $O/pnp_pmc.log 2>&1; tail -20 $O/pnp_pmc.log ;;
    eigen) timeout 300 python scripts/eigen_pin.py 2>&1 | grep -E "eigen_found|directories_walked|bit_identical" ;;
    icp) timeout 200 python scripts/gpu_icp_perf.py 2>&1 | grep ICP | tee $O/icp_perf.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
