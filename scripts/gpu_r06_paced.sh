# Where does the launched tick's 10 Hz penalty come from?  The C caller at 10k rows with the pause SLEPT through (the reference's rate.sleep)
# and SPUN through (same GPU idle time, CPU core awake), launched and resident, over pause lengths; twice, alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T=cerebro_amd/lib/sync_tick_latency
(for rep in 1 2; do
 for ms in 0 1 10 100 300; do
  for mode in sleep spin; do
   [ $ms = 0 ] && [ $mode = spin ] && continue
   for res in 0 1; do
     n=40; [ $ms = 0 ] && n=400
     echo -n "rep $rep pause_ms $ms $mode resident $res: "
     CHIP_TICK_RESIDENT=$res $T 10000 $n 0 $ms $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sync_tick']; e=d['enqueue']; c=d['collect']
print(f\"tick mean {s['mean_us']:6.1f} p50 {s['p50_us']:6.1f} min {s['min_us']:6.1f} | enqueue p50 {e['p50_us']:5.1f} collect p50 {c['p50_us']:5.1f}\")"
   done
  done
 done
done) | tee gpurun_out/r06/paced_sleep_vs_spin.txt
