# kernel durations of the PnP pair by stage (CHIP_PNP_DEBUG_STOP = 1 entry only, 2 after Hessenberg, 3 after the QR iteration, 4 after the
# back-substitution, 0 full) for H hypotheses (default 50 = reference mode), rocprofv3 kernel trace, 20 calls each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H=${1:-50}
O=gpurun_out/r04
mkdir -p $O
for stop in 1 2 3 4 0; do
  rm -rf $O/pnp_stage_$stop
  CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=cerebro_amd/lib/hooks/libcerebro_hip.so CHIP_PNP_DEBUG_STOP=$stop timeout 300 rocprofv3 --kernel-trace -d $O/pnp_stage_$stop -o r -- python scripts/run_pnp_ref_mode.py $H > $O/pnp_stage_$stop.log 2>&1
  python - <<PY
import sqlite3,glob
for db in glob.glob("$O/pnp_stage_$stop/**/*_results.db", recursive=True):
    con=sqlite3.connect(db)
    for n,c,a,mn,mx in con.execute("select name,count(*),avg(duration)/1e3,min(duration)/1e3,max(duration)/1e3 from kernels where name like '%pnp_%' group by name"):
        print("H=$H stop=$stop", n[:40], c, f"avg {a:.1f} min {mn:.1f} max {mx:.1f} us")
PY
done | tee $O/pnp_stages_$H.txt
