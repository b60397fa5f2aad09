# Round-2 evidence run C (final structure): full GPU suite, default bench, kernel-trace stats of the same command, PMC passes
# (HBM traffic of db_scan_topk and db_gemm_topk: separate runs per counter), MFMA-kernel counters, multi-GPU host-cost probe,
# scan-reserve experiment on the plain ctx.
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1; echo exit=$? >> gpurun_out/r02/pytest_gpu.log)
grep -E "passed|failed|exit=" gpurun_out/r02/pytest_gpu.log | tail -3
(timeout 900 python bench.py > gpurun_out/r02/bench_default.log 2>&1; echo exit=$? >> gpurun_out/r02/bench_default.log)
tail -2 gpurun_out/r02/bench_default.log | cut -c1-400
rm -rf gpurun_out/r02/trace gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r02/trace -o r02 -- python bench.py --cpu-budget 0 > gpurun_out/r02/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02/pmc_fetch -o r02 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-sizes > gpurun_out/r02/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r02/pmc_write -o r02 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-sizes > gpurun_out/r02/pmc_write.log 2>&1
bash scripts/gpu_batch_pmc2.sh > gpurun_out/r02/batch_pmc.txt 2>&1
grep -E "^(SQ_|GRBM|FETCH)" gpurun_out/r02/batch_pmc.txt | head -30
(timeout 600 python scripts/gpu_group_cost.py > gpurun_out/r02/group_cost.json 2>gpurun_out/r02/group_cost.err; echo exit=$?)
(CHIP_SCAN_RESERVE=4 timeout 600 python bench.py --cpu-budget 0 --no-pnp --no-batch > gpurun_out/r02/bench_reserve4.log 2>&1; echo exit=$?)
(timeout 600 python bench.py --cpu-budget 0 --no-pnp --no-batch --storage f64 --no-sizes > gpurun_out/r02/bench_f64.log 2>&1; echo exit=$?)
(timeout 600 python bench.py --cpu-budget 0 --no-pnp --no-batch --force-sharded --rows 125000 > gpurun_out/r02/bench_sharded_125k.log 2>&1; echo exit=$?)
(timeout 600 python bench.py --cpu-budget 0 --no-pnp --no-batch --gpus 8 --same-device > gpurun_out/r02/bench_group8_same_device.log 2>&1; echo exit=$?)
python scripts/gpu_batch_variants.py 2>&1 | grep "KC=" > gpurun_out/r02/batch_variants.txt; cat gpurun_out/r02/batch_variants.txt
ls gpurun_out/r02
