#!/usr/bin/env bash
# scripts/preflight_8gpu.sh [N=8] [bench args ...] -- what to run FIRST on a multi-GPU node.
# The driver's 8-GPU run is the first execution of real RCCL at N > 1 (real RCCL refuses two ranks on one device, so the 1-GPU boxes
# of the build pool can only run the exchange over the shared-memory stand-in of tests/fakerccl).  Three tests of tests/test_multi_gpu.py
# are gated on the device count and have therefore never run; this script runs exactly those first, then the benchmark in the driver's
# launch shape, and puts their verdict into the bench line (config.preflight) so that the scaling number says what it stands on:
#   test_group_over_rccl_distinct_devices            one process, G devices: ncclCommInitAll + in-stream ncclAllGather over xGMI
#   test_comm_init_rank_real_processes_over_rccl     one process per GPU: chip_comm_init_rank + ncclAllGather / ncclBroadcast
#   test_group_copy_exchange_distinct_devices        the fallback exchange: hipMemcpyPeerAsync between distinct devices
# Exit code: that of the benchmark (the preflight's verdict is data, not a gate -- a failed preflight with a working fallback exchange
# still yields a JSON line, whose config.exchange / exchange_fallback say which exchange carried it).
set -u
cd "$(dirname "$0")/.."
N=${1:-8}
shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "[preflight] $NDEV device(s) visible, benchmark at N=$N" >&2
if [ "$NDEV" -ge 2 ]; then
  OUT=$(timeout 1500 python -m pytest tests/test_multi_gpu.py -q -m gpu -p no:cacheprovider \
        -k "test_group_over_rccl_distinct_devices or test_comm_init_rank_real_processes_over_rccl or test_group_copy_exchange_distinct_devices" 2>&1 | tail -3)
  VERDICT=$(echo "$OUT" | grep -Eo "[0-9]+ (passed|failed|skipped|error|errors)[^=]*" | tail -1)
  export BENCH_PREFLIGHT="device-count-gated tests on $NDEV devices: ${VERDICT:-no verdict (pytest did not finish)}"
else
  export BENCH_PREFLIGHT="not run: $NDEV device visible (the three device-count-gated tests need >= 2)"
fi
echo "[preflight] $BENCH_PREFLIGHT" >&2
if [ "$N" -gt 1 ]; then
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29533}" \
       bench.py --gpus "$N" "$@"
else
  exec python bench.py --gpus 1 "$@"
fi
