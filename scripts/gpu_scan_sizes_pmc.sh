# HBM-side traffic + isolated kernel durations of the size legs' launch shapes: BASELINE configs 2, 3 (10k / 29k / 100k rows x 4096-D float rows)
# and the reference's two production shapes (8192-D x 29k float rows, Cerebro.cpp:946,1021; 4096-D x 1M DOUBLE rows, server.py:148-149):
# synchronous ticks (one launch at a time) under rocprofv3, FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace only
# (MI355X_MICROARCH.md, HBM section) -> profiles/scan_traffic_sizes.json + profiles/r06_sizes_pmc.md via scripts/summarize_sizes_pmc.py
OUT=gpurun_out/sizes_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT/*
for shape in "10000 4096 f32" "29000 4096 f32" "100000 4096 f32" "29000 8192 f32" "1000000 4096 f64"; do
  set -- $shape
  rows=$1; dim=$2; st=$3; tag=${rows}_${dim}_${st}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/${tag}_$ctr -o p -- python scripts/run_ticks_once.py $rows sync 60 $dim $st > $OUT/${tag}_$ctr.log 2>&1
    tail -1 $OUT/${tag}_$ctr.log
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${tag}_trace -o p -- python scripts/run_ticks_once.py $rows sync 60 $dim $st > $OUT/${tag}_trace.log 2>&1
done
python scripts/summarize_sizes_pmc.py $OUT
