# MFMA many-query kernel (db_gemm_topk): stall breakdown from SQ counters, a few per pass (separate runs; no trace domains beyond --kernel-trace)
mkdir -p gpurun_out/r02/batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|GRBM_GUI_ACTIVE|TCP_[A-Z_0-9]+|FETCH_SIZE|MfmaUtil|LdsBankConflict" | sort -u > gpurun_out/r02/batch/counters.txt
wc -l gpurun_out/r02/batch/counters.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/r02/batch/pmc$i -o b -- python scripts/run_batch_once.py > gpurun_out/r02/batch/pmc$i.log 2>&1
  tail -1 gpurun_out/r02/batch/pmc$i.log | cut -c1-200
done
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('gpurun_out/r02/batch/pmc*/*_results.db')):
    con = sqlite3.connect(db)
    try:
        for n, cn, c, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%db_gemm%' group by kernel_name, counter_name"):
            print(f"{cn:32s} {c:3d} {a:.5e}")
    except Exception as e:
        print(db, e)
PY
