for rep in 1 2 3; do for n in 1 2; do export CHIP_SCAN_STREAMS=$n; echo -n "streams $n: "
python scripts/gpu_tick_host_times.py 1000000 16 2>&1 | grep rows=
done; done
