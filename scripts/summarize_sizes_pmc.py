#!/usr/bin/env python3
"""gpurun_out/sizes_pmc/<rows>_<dim>_<f32|f64>_{FETCH_SIZE,WRITE_SIZE,trace}/*_results.db -> profiles/scan_traffic_sizes.json + profiles/r06_sizes_pmc.md.
Per shape: the scan kernel of a SYNCHRONOUS tick (one launch at a time): launches, avg / min duration (kernel trace), FETCH_SIZE and
WRITE_SIZE per launch (separate passes), traffic = FETCH_SIZE x 1024 x 2 (gfx950: 128-B requests tallied at 64 B for 16-B-per-lane
streaming reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x 1024.  Only the steady-state launches count: the warm-up ticks of
the driver script are pipelined, so the LAST `n_sync` launches of the scan kernel are the synchronous ones.
"sizes" = the BASELINE shape (4096-D float rows) keyed by rows; "shapes" = the reference's production shapes keyed by bench.shape_name."""
import glob
import json
import sqlite3
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (shape_name only)

N_SYNC = 60
SHAPES = [(10000, 4096, "f32"), (29000, 4096, "f32"), (100000, 4096, "f32"), (29000, 8192, "f32"), (1000000, 4096, "f64")]


def last_sync(con, table, cols, where_name):
    rows = con.execute(f"select {cols} from {table} where {where_name} like '%db_scan_topk%' order by start").fetchall()
    return rows[-N_SYNC:]


def main(src):
    src = ROOT / src
    out = {"tag": "r06", "correction": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section); FETCH_SIZE counts the "
           "L2's fabric-side read requests, Infinity-Cache hits included -- for a cache-sized prefix it is NOT DRAM bytes",
           "mode": f"synchronous ticks (chip_loop_tick), the last {N_SYNC} scan launches of each pass", "sizes": {}, "shapes": {}}
    md = ["# size legs: per-launch counters and durations of the scan kernel (synchronous ticks, one launch at a time)", "",
          "`algorithmic` = 4*D*rows (SURVEY 8d: priced on the fp32 layout whatever the storage type); `actual` = the bytes of the stored prefix "
          "(8*D*rows for double rows).", "",
          "| shape | kernel | launches | avg (us) | min (us) | algorithmic bytes | actual bytes | FETCH_SIZE avg (KiB) | WRITE_SIZE avg (KiB) | traffic (B) | traffic / actual | frac_kernel (avg, algorithmic) | frac_kernel (min, algorithmic) | frac (avg, actual bytes) |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for rows, dim, st in SHAPES:
        tag = f"{rows}_{dim}_{st}"
        e = {"db_rows": rows, "D": dim, "storage": st, "algorithmic_bytes_per_launch": 4.0 * dim * rows,
             "actual_bytes_per_launch": (8.0 if st == "f64" else 4.0) * dim * rows}
        for db in glob.glob(str(src / f"{tag}_trace" / "**" / "*_results.db"), recursive=True):
            con = sqlite3.connect(db)
            r = last_sync(con, "kernels", "name, duration", "name")
            if r:
                d = [x[1] / 1e3 for x in r]
                e.update(kernel=r[-1][0], launches=len(d), avg_kernel_us=sum(d) / len(d), min_kernel_us=min(d), max_kernel_us=max(d))
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            for db in glob.glob(str(src / f"{tag}_{ctr}" / "**" / "*_results.db"), recursive=True):
                con = sqlite3.connect(db)
                r = con.execute("select value from counters_collection where kernel_name like '%db_scan_topk%' and counter_name=? order by start", (ctr,)).fetchall()[-N_SYNC:]
                if r:
                    v = [x[0] for x in r]
                    e[ctr.lower() + "_kib_avg"] = sum(v) / len(v)
                    e[ctr.lower() + "_kib_min"] = min(v)
                    e[ctr.lower() + "_kib_max"] = max(v)
        if "fetch_size_kib_avg" in e:
            e["hbm_bytes_per_launch"] = e["fetch_size_kib_avg"] * 1024 * 2 + e.get("write_size_kib_avg", 0.0) * 1024
        if dim == 4096 and st == "f32":
            out["sizes"][str(rows)] = e
        else:
            out["shapes"][bench.shape_name(rows, dim, st)] = e
        alg, act = e["algorithmic_bytes_per_launch"], e["actual_bytes_per_launch"]
        if "avg_kernel_us" in e and "hbm_bytes_per_launch" in e:
            md.append(f"| {bench.shape_name(rows, dim, st)} | `{e['kernel'][:60]}` | {e['launches']} | {e['avg_kernel_us']:.2f} | {e['min_kernel_us']:.2f} | {alg:.4e} | {act:.4e} | "
                      f"{e['fetch_size_kib_avg']:.1f} | {e.get('write_size_kib_avg', 0.0):.1f} | {e['hbm_bytes_per_launch']:.4e} | "
                      f"{e['hbm_bytes_per_launch'] / act:.4f} | {alg / (e['avg_kernel_us'] * 1e-6) / 8e12:.3f} | {alg / (e['min_kernel_us'] * 1e-6) / 8e12:.3f} | "
                      f"{act / (e['avg_kernel_us'] * 1e-6) / 8e12:.3f} |")
    (ROOT / "profiles" / "scan_traffic_sizes.json").write_text(json.dumps(out, indent=1))
    (ROOT / "profiles" / "r06_sizes_pmc.md").write_text("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sizes_pmc")
