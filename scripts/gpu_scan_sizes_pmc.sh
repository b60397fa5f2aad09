# HBM-side traffic + isolated kernel durations of the size legs' launch shapes (BASELINE configs 2, 3: 10k / 29k / 100k rows x 4096-D):
# synchronous ticks (one launch at a time) under rocprofv3, FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace only
# (MI355X_MICROARCH.md, HBM section) -> profiles/scan_traffic_sizes.json via scripts/summarize_sizes_pmc.py
OUT=gpurun_out/sizes_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT/*
for rows in 10000 29000 100000; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/${rows}_$ctr -o p -- python scripts/run_ticks_once.py $rows sync 60 > $OUT/${rows}_$ctr.log 2>&1
    tail -1 $OUT/${rows}_$ctr.log
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${rows}_trace -o p -- python scripts/run_ticks_once.py $rows sync 60 > $OUT/${rows}_trace.log 2>&1
done
python scripts/summarize_sizes_pmc.py $OUT
