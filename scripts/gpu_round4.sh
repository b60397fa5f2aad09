# round-4 GPU calls, one parameterised script:  bash scripts/gpu_round4.sh <step> [<step> ...]
#   suite      pytest -m gpu (whole parity suite)             -> gpurun_out/r04/pytest_gpu.log
#   bench      default python bench.py                        -> gpurun_out/r04/bench_default.log
#   trace      rocprofv3 --kernel-trace --stats of the same   -> gpurun_out/r04/trace (summarised by scripts/summarize_rocprof.py)
#   scanpmc    FETCH_SIZE / WRITE_SIZE passes of the 1M tick  -> gpurun_out/r04/pmc_*
#   sizespmc   scripts/gpu_scan_sizes_pmc.sh                  -> profiles/scan_traffic_sizes.json (copied back by hand)
#   pnp        PnP parity tests + fuzz + call rates + stamps  -> gpurun_out/r04/pnp_*.txt
#   pnppmc     SQ counters of the PnP kernel pair             -> gpurun_out/r04/pnp_pmc
#   tests:<expr>   pytest -m gpu -k <expr>
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
for step in "$@"; do
  case $step in
    suite) (timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo pytest_exit=$? >> $O/pytest_gpu.log); tail -15 $O/pytest_gpu.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl" ;;
    tests:*) (timeout 1500 python -m pytest tests -m gpu -q -x -k "${step#tests:}" > $O/pytest_k.log 2>&1; echo pytest_exit=$? >> $O/pytest_k.log); tail -25 $O/pytest_k.log | grep -vE "RCCL|HIP ver|ROCm ver|Hostname|Librccl" ;;
    bench) (timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo exit=$? >> $O/bench_default.log); tail -2 $O/bench_default.log | cut -c1-1500; tail -5 $O/bench_default.err ;;
    trace) rm -rf $O/trace; timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r04 -- python bench.py --cpu-budget 0 > $O/trace.log 2>&1; grep '^{' $O/trace.log | cut -c1-200 ;;
    scanpmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o r04 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > $O/pmc_$c.log 2>&1; done ;;
    sizespmc) bash scripts/gpu_scan_sizes_pmc.sh ;;
    pnp) (timeout 900 python -m pytest tests -m gpu -q -x -k "pnp or config3 or fuzz or golden or consistency" > $O/pytest_pnp.log 2>&1; echo pytest_exit=$? >> $O/pytest_pnp.log); tail -4 $O/pytest_pnp.log
         timeout 600 python scripts/gpu_pnp_fuzz.py > $O/pnp_fuzz.txt 2>&1; tail -2 $O/pnp_fuzz.txt
         for i in 1 2 3; do timeout 300 python scripts/gpu_pnp_rates.py 2>&1 | tail -1; done | tee $O/pnp_rates.txt
         timeout 300 python scripts/gpu_pnp_stamps.py 50 > $O/pnp_stamps.txt 2>&1; tail -30 $O/pnp_stamps.txt ;;
    pnppmc) bash scripts/gpu_pnp_pmc.sh ;;
    *) echo "unknown step $step" ;;
  esac
done
