# db_scan_topk (4096-D x 1M ticks): SQ counters, a few per pass (separate runs; no trace domains beyond --kernel-trace)
mkdir -p gpurun_out/r02/scan_pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/r02/scan_pmc/pmc$i -o s -- python bench.py --steps 10 --warmup 2 --cpu-budget 0 --no-pnp --no-batch --no-sizes > gpurun_out/r02/scan_pmc/pmc$i.log 2>&1
  tail -1 gpurun_out/r02/scan_pmc/pmc$i.log | cut -c1-120
done
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('gpurun_out/r02/scan_pmc/pmc*/*_results.db')):
    con = sqlite3.connect(db)
    try:
        for n, cn, c, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%db_scan_topk%' group by kernel_name, counter_name"):
            print(f"{cn:32s} {c:3d} {a:.5e}")
    except Exception as e:
        print(db, e)
PY
