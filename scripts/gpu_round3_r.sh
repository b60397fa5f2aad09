# round 3, call R: kernel durations of the PnP pair in reference mode by stage (rocprofv3 kernel trace; CHIP_PNP_DEBUG_STOP = 3, 4, 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for stop in 3 4 0; do
  rm -rf gpurun_out/prof_r_$stop
  CHIP_PNP_DEBUG_STOP=$stop timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r_$stop -o r -- python scripts/run_pnp_ref_mode.py 50 > gpurun_out/prof_r_$stop.log 2>&1
  python - <<PY
import sqlite3,glob
for db in glob.glob("gpurun_out/prof_r_$stop/*_results.db"):
    con=sqlite3.connect(db)
    for n,c,a,mn,mx in con.execute("select name,count(*),avg(duration)/1e3,min(duration)/1e3,max(duration)/1e3 from kernels group by name"):
        print("stop=$stop", n[:50], c, f"avg {a:.1f} min {mn:.1f} max {mx:.1f} us")
PY
done
