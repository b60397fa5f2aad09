# round-6 evidence run: whole GPU suite, default bench, rocprofv3 trace of the driver's command, PMC passes (1M scan, size / shape legs, batched PnP)
bash scripts/gpu_round6.sh suite bench trace scanpmc sizespmc pnppmc
cd $GRAFT_REPO_ROOT && python scripts/summarize_rocprof.py r06 gpurun_out/r06/prof 2>&1 | tail -30
