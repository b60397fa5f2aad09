mkdir -p gpurun_out/shtrace; rm -rf gpurun_out/shtrace/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for prio in 0 1; do
export CHIP_SCAN_STREAM_PRIORITY=$prio
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/shtrace/p$prio -o sh -- python scripts/gpu_sharded_variants.py 125000 > gpurun_out/shtrace/log$prio.txt 2>&1
grep rows= gpurun_out/shtrace/log$prio.txt
python - <<PY
import sqlite3, glob
db = glob.glob('gpurun_out/shtrace/p$prio/*.db')[0]
con = sqlite3.connect(db)
print(con.execute("select name, queue_id, stream_id, count(*) from kernels group by name, queue_id, stream_id order by stream_id").fetchall())
PY
done
