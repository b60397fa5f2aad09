# Where does a reference-mode PnP call (<= 50 hypotheses, the production path) spend its time?  (1) host phases of the call
# (CHIP_PNP_HOST_TIMING=1, no profiler); (2) rocprofv3 kernel + memory-copy trace of 20 calls: start / end of the H2D copy, pnp_build_solve,
# pnp_eig_score per call -> gaps between them and from call start to first kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for H in 0 50 1000; do
  CHIP_PNP_HOST_TIMING=1 timeout 300 python scripts/run_pnp_ref_mode.py $H 2>&1 | grep -E "pnp host timing|ok" | sed "s/^/H=$H /"
done | tee $O/pnp_host_timing.txt
rm -rf $O/pnp_tl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/pnp_tl -o t -- python scripts/run_pnp_ref_mode.py 50 > $O/pnp_tl.log 2>&1
python - <<'PY' | tee gpurun_out/r04/pnp_timeline.txt
import sqlite3, glob
for db in glob.glob("gpurun_out/r04/pnp_tl/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    ev = [(s, e, n[:28]) for n, s, e in con.execute("select name, start, end from kernels where name like '%pnp_%'")]
    try:
        ev += [(s, e, "memcpy " + str(n)) for n, s, e in con.execute("select name, start, end from memory_copies")]
    except Exception as ex:
        print("no memory_copies view:", ex)
    ev.sort()
    builds = [x for x in ev if "build" in x[2]]
    print(f"{len(builds)} calls")
    rows = []
    for b in builds[5:]:
        eig = min((x for x in ev if "eig" in x[2] and x[0] >= b[1]), key=lambda x: x[0])
        cp = max((x for x in ev if x[2].startswith("memcpy") and x[1] <= b[0]), key=lambda x: x[1], default=None)
        rows.append(((cp[1] - cp[0]) / 1e3 if cp else 0, (b[0] - cp[1]) / 1e3 if cp else 0, (b[1] - b[0]) / 1e3, (eig[0] - b[1]) / 1e3, (eig[1] - eig[0]) / 1e3))
    import statistics as st
    names = ["H2D copy", "copy end -> build start", "pnp_build_solve", "build end -> eig start", "pnp_eig_score"]
    for i, n in enumerate(names):
        v = [r[i] for r in rows]
        print(f"{n:28s} mean {st.mean(v):8.1f} us   min {min(v):8.1f}   max {max(v):8.1f}")
    print(f"{'sum':28s} mean {st.mean([sum(r) for r in rows]):8.1f} us")
PY
