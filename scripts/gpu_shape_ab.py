"""One production shape of the scan under the current CHIP_SCAN_* environment (one process per configuration):
  python scripts/gpu_shape_ab.py ROWS DIM [f32|f64]   -> isolated kernel us / frac, pipelined step us, synchronous tick us"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
rows, dim = int(sys.argv[1]), int(sys.argv[2])
st = sys.argv[3] if len(sys.argv) > 3 else "f32"
leg = bench.shape_leg(rows, dim, st, 16, n_ticks=120)
r = leg["roofline"]
print(json.dumps({"shape": bench.shape_name(rows, dim, st), "kernel_us": round(r["isolated_kernel_ms"] * 1e3, 2), "frac_kernel": round(r["frac_kernel"], 4),
                  "step_us": round(leg["ms_per_step"] * 1e3, 2), "frac_step": round(r["frac_step"], 4), "sync_tick_us": round(leg["sync_tick_us"], 2),
                  "kernel": r["kernel"][:24]}))
