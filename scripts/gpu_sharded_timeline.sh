# kernel timeline of the sharded tick path (world 1, RCCL): shows scans back to back on the two scan streams while the local
# merge / exchange copy / global merge of the previous ticks run on the ctx stream
mkdir -p gpurun_out/shtl; rm -rf gpurun_out/shtl/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/shtl -o tl -- python bench.py --rows ${1:-500000} --steps 60 --warmup 10 --cpu-budget 0 --no-pnp --no-batch --force-sharded > gpurun_out/shtl/log.txt 2>&1
grep '^{' gpurun_out/shtl/log.txt | cut -c1-200
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob('gpurun_out/shtl/*.db')[0])
rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
n = len(rows); i0 = n // 2
base = rows[i0][1]
print("| start (us) | end (us) | dur (us) | queue | stream | kernel |"); print("|---|---|---|---|---|---|")
for name, s, e, q, st in rows[i0:i0 + 20]:
    print(f"| {(s-base)/1e3:.1f} | {(e-base)/1e3:.1f} | {(e-s)/1e3:.1f} | {q} | {st} | `{name[:48]}` |")
PY
