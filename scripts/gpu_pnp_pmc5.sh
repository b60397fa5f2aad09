# Counters + durations of the BATCHED PnP call (8 problems x 1000 hypotheses per launch pair, 5 calls) -> gpurun_out/r06/pnp_pmc.json
# (copied to profiles/pnp_pmc.json: bench.py's pnp.roofline.valu_busy).  Separate passes: kernel trace, then --pmc (no other trace domain).
O=gpurun_out/r06/pnp_pmc
mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o p -- python scripts/run_pnp_batch_once.py > $O/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_valu -o p -- python scripts/run_pnp_batch_once.py > $O/pmc_valu.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace -d $O/pmc_lds -o p -- python scripts/run_pnp_batch_once.py > $O/pmc_lds.log 2>&1
python - <<'PY'
import sqlite3, glob, json
O = "gpurun_out/r06/pnp_pmc"
dur, ctr = {}, {}
def short(n):
    return "pnp_build_solve" if "pnp_build_solve" in n else "pnp_eig_score" if "pnp_eig_score" in n else n
for db in glob.glob(O + "/trace/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    for n, c, a, mn, mx in con.execute("select name,count(*),avg(duration)/1e3,min(duration)/1e3,max(duration)/1e3 from kernels where name like '%pnp_%' group by name"):
        dur[short(n)] = {"calls": c, "avg_us": a, "min_us": mn, "max_us": mx}
for pas in ("pmc_valu", "pmc_lds"):
    for db in glob.glob(O + f"/{pas}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db)
        for n, cn, c, a in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%pnp_%' group by kernel_name, counter_name"):
            ctr.setdefault(short(n), {})[cn] = a
SIMDS, CLK = 1024, 2.4e9
busy = {k: ctr[k]["SQ_INSTS_VALU"] * 4.0 / (SIMDS * dur[k]["avg_us"] * 1e-6 * CLK) for k in dur if "SQ_INSTS_VALU" in ctr.get(k, {})}
out = {"workload": "chip_pnp_ransac_batch: 8 problems x 1000 hypotheses x 512 correspondences per launch pair (scripts/run_pnp_batch_once.py, 5 calls)",
       "source": "profiles/pnp_pmc.json: rocprofv3 --pmc SQ_INSTS_VALU ... (separate pass) + --kernel-trace durations; valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel duration x 2.4 GHz)",
       "valu_busy": busy, "kernels": dur, "counters_avg_per_launch": ctr,
       "valu_insts_per_hypothesis": {k: ctr[k]["SQ_INSTS_VALU"] / 8000.0 for k in ctr if "SQ_INSTS_VALU" in ctr[k]}}
open("gpurun_out/r06/pnp_pmc.json", "w").write(json.dumps(out, indent=1) + "\n")
print(json.dumps({"valu_busy": busy, "kernels": dur}, indent=1))
PY
