mkdir -p gpurun_out/prof_pnp; rm -rf gpurun_out/prof_pnp/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -E "SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_INSTS_LDS|SQ_WAIT_INST_LDS|SQ_INST_CYCLES_VMEM|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_SALU|SQ_ACTIVE_INST_LDS|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_THREAD_CYCLES_VALU|SQ_LDS_BANK_CONFLICT" | cut -c1-120 | sort -u | head -40 > gpurun_out/prof_pnp/counters.txt
cat gpurun_out/prof_pnp/counters.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pnp/trace -o r01 -- python scripts/run_pnp_once.py > gpurun_out/prof_pnp/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_pnp/pmc_valu -o r01 -- python scripts/run_pnp_once.py > gpurun_out/prof_pnp/pmc_valu.log 2>&1
tail -1 gpurun_out/prof_pnp/pmc_valu.log
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d gpurun_out/prof_pnp/pmc_lds -o r01 -- python scripts/run_pnp_once.py > gpurun_out/prof_pnp/pmc_lds.log 2>&1
tail -1 gpurun_out/prof_pnp/pmc_lds.log
ls gpurun_out/prof_pnp/*/
