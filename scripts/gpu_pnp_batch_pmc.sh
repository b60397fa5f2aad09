# SQ counters + kernel durations of the BATCHED PnP call (8 problems x 1000 hypotheses per launch pair, 5 calls): what bounds the batched rate
O=gpurun_out/r04/pnp_batch_pmc
mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o p -- python scripts/run_pnp_batch_once.py > $O/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu -o p -- python scripts/run_pnp_batch_once.py > $O/pmc_valu.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d $O/pmc_lds -o p -- python scripts/run_pnp_batch_once.py > $O/pmc_lds.log 2>&1
python - <<'PY' | tee gpurun_out/r04/pnp_batch_pmc.txt
import sqlite3, glob
O = "gpurun_out/r04/pnp_batch_pmc"
for db in glob.glob(O + "/trace/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for n, c, a, mn, mx in con.execute("select name,count(*),avg(duration)/1e3,min(duration)/1e3,max(duration)/1e3 from kernels where name like '%pnp_%' group by name"):
        print(f"{n[:44]:44s} calls {c}  avg {a:.1f} us  min {mn:.1f}  max {mx:.1f}")
for pas in ("pmc_valu", "pmc_lds"):
    for db in glob.glob(O + f"/{pas}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        for n, cn, c, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%pnp_%' group by kernel_name, counter_name"):
            print(f"{n[:44]:44s} {cn:24s} {c:3d} dispatches  avg {a:.4e}")
PY
