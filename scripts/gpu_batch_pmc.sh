mkdir -p gpurun_out/prof_batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|VALUBusy|MfmaUtil" | head -30 > gpurun_out/prof_batch/counters.txt
cat gpurun_out/prof_batch/counters.txt | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_batch/trace -o r01 -- python scripts/run_batch_once.py > gpurun_out/prof_batch/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_batch/pmc_mfma -o r01 -- python scripts/run_batch_once.py > gpurun_out/prof_batch/pmc_mfma.log 2>&1
tail -2 gpurun_out/prof_batch/pmc_mfma.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_batch/pmc_fetch -o r01 -- python scripts/run_batch_once.py > gpurun_out/prof_batch/pmc_fetch.log 2>&1
ls gpurun_out/prof_batch/*
