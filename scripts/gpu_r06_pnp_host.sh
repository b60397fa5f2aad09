# new PnP / unit-data GPU tests, then the host-side split of a reference-mode PnP call (CHIP_PNP_HOST_TIMING) and the rocprofv3 timeline of its two kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_unit_data_gpu.py tests/test_pnp_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06/pytest_unit_pnp.log 2>&1; tail -3 gpurun_out/r06/pytest_unit_pnp.log
(for i in 1 2 3; do CHIP_PNP_HOST_TIMING=1 python scripts/run_pnp_ref_mode.py 0 2>&1 | grep -v "amdgpu.ids" | tail -2; done) | tee gpurun_out/r06/pnp_host_timing.txt
python scripts/gpu_pnp_rates.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/pnp_host_timing.txt
rm -rf gpurun_out/r06/prof/pnp_ref; timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r06/prof/pnp_ref -o t -- python scripts/run_pnp_ref_mode.py 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06/prof/pnp_ref/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows = [r for r in rows if 'pnp_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
b = [r for r in rows if 'build' in r['Kernel_Name']]; e = [r for r in rows if 'eig' in r['Kernel_Name']]
import statistics as st
db = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in b]
de = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in e]
gap = [(int(y['Start_Timestamp']) - int(x['End_Timestamp'])) / 1e3 for x, y in zip(b, e)]
print(f"pnp_build_solve {st.mean(db):.1f} us  gap {st.mean(gap):.1f} us  pnp_eig_score {st.mean(de):.1f} us (min {min(de):.1f} max {max(de):.1f})  sum {st.mean(db)+st.mean(gap)+st.mean(de):.1f}")
PY
