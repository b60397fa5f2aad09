mkdir -p gpurun_out/prof_tick; rm -rf gpurun_out/prof_tick/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_tick/t10k -o t -- python scripts/run_ticks_once.py 10000 > gpurun_out/prof_tick/t10k.log 2>&1; tail -2 gpurun_out/prof_tick/t10k.log
CHIP_TICK_SAME_STREAM=0 CHIP_SCAN_ROWS=-1 CHIP_SCAN_SHORT_BPC=0 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_tick/t10k_r2 -o t -- python scripts/run_ticks_once.py 10000 > gpurun_out/prof_tick/t10k_r2.log 2>&1; tail -1 gpurun_out/prof_tick/t10k_r2.log
python - <<'PY'
import sqlite3, glob
for d in ("t10k", "t10k_r2"):
    db = glob.glob(f"gpurun_out/prof_tick/{d}/*_results.db")[0]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    rows = [r for r in rows if "scan_topk" in r[0] or "topk_merge" in r[0]]
    rows = rows[len(rows)//2: len(rows)//2 + 24]
    t0 = rows[0][1]
    print(d)
    for n, s, e, q, st in rows:
        print(f"  {'scan ' if 'scan' in n else 'merge'} q{q} s{st}  start {(s-t0)/1e3:8.2f}  end {(e-t0)/1e3:8.2f}  dur {(e-s)/1e3:6.2f}")
PY
