for prio in 0 1 -1; do for q in "" 8 16; do
  echo "== scan stream priority=$prio GPU_MAX_HW_QUEUES=${q:-default}"
  ( [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; export CHIP_SCAN_STREAM_PRIORITY=$prio
    python scripts/gpu_sharded_variants.py 125000 2>&1 | grep rows= )
done; done
