# unit-L2 synthetic data (ABI 7): the new GPU tests, then the driver-shaped bench on it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_unit_data_gpu.py tests/test_capi_symbols.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06/pytest_unit.log 2>&1; tail -5 gpurun_out/r06/pytest_unit.log
timeout 900 python bench.py > gpurun_out/r06/bench_unit.log 2> gpurun_out/r06/bench_unit.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r06/bench_unit.log; tail -5 gpurun_out/r06/bench_unit.err
