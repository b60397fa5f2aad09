# Round-2 evidence run B: multi-GPU host-cost probe + PMC passes (HBM traffic of db_scan_topk; separate runs per counter)
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python scripts/gpu_group_cost.py > gpurun_out/r02/group_cost.json 2>gpurun_out/r02/group_cost.err; echo exit=$?)
cat gpurun_out/r02/group_cost.json | head -70
rm -rf gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02/pmc_fetch -o r02 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/r02/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r02/pmc_write -o r02 -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/r02/pmc_write.log 2>&1
ls gpurun_out/r02/*/
