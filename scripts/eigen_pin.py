#!/usr/bin/env python3
"""Pin what can be pinned: look for a real <Eigen/Dense> EVERYWHERE a machine might keep one -- EIGEN3_INCLUDE_DIR,
the system include directories, /opt, every Python site-packages / dist-packages on sys.path and under the interpreter prefixes (torch,
tensorflow, pybind11 and conda ship headers there), conda prefixes -- and, when one is found, build oracle/eigen_probe.cc against it with
the reference's Release flags, run the LITERAL statements of Cerebro.cpp:1026-1028 on a 10k-column fixture and compare u, um, umm with
the Eigen-order restatement (orc_ref_scan_f64_eigen_gemv3) bit for bit.  Test infrastructure; never on the product path.
Writes gpurun_out/r05/eigen_pin.json (copied to profiles/r05_eigen_pin.json): found / not found, where it looked, the comparison."""
import glob
import json
import os
import shutil
import site
import subprocess
import sys
import sysconfig
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def find_eigen():
    roots = [os.environ.get("EIGEN3_INCLUDE_DIR", ""), "/usr/include/eigen3", "/usr/local/include/eigen3", "/usr/include", "/usr/local/include", "/opt/eigen3"]
    roots += glob.glob("/opt/*/include/eigen3") + glob.glob("/opt/*/include") + glob.glob("/opt/conda*/include/eigen3") + glob.glob(os.path.expanduser("~/*conda*/include/eigen3"))
    py = set(p for p in sys.path if p and os.path.isdir(p))
    try:
        py.update(site.getsitepackages()); py.add(site.getusersitepackages())
    except Exception:
        pass
    for k in ("purelib", "platlib", "include", "platinclude", "data"):
        v = sysconfig.get_paths().get(k)
        if v:
            py.add(v)
    scanned, hits = 0, []
    for r in roots:
        if r and os.path.exists(os.path.join(r, "Eigen", "Dense")):
            hits.append(r)
    for base in sorted(py):
        if not os.path.isdir(base):
            continue
        for dirpath, dirnames, filenames in os.walk(base):
            scanned += 1
            if dirpath.count(os.sep) - base.count(os.sep) > 6:
                dirnames[:] = []
                continue
            if os.path.basename(dirpath) == "Eigen" and "Dense" in filenames:
                hits.append(os.path.dirname(dirpath))
    return hits, sorted(r for r in roots if r), sorted(py), scanned


def main():
    out_dir = ROOT / "gpurun_out" / "r05"
    out_dir.mkdir(parents=True, exist_ok=True)
    hits, roots, pyroots, scanned = find_eigen()
    rep = {"eigen_found": bool(hits), "include_dirs_found": hits, "searched_fixed": roots, "searched_python_roots": pyroots,
           "directories_walked": scanned, "host": os.uname().nodename}
    gxx = shutil.which("g++")
    if hits and gxx:
        import numpy as np
        import oracle_lib
        D, k = 4096, 10_000
        M = oracle_lib.synth_rows(20190412, range(k + 3), D).astype(np.float64)        # rows = the columns of the reference's M
        fin, fout, exe = out_dir / "eigen_in.bin", out_dir / "eigen_out.bin", out_dir / "eigen_probe"
        with open(fin, "wb") as f:
            f.write(np.array([D, k], dtype=np.int32).tobytes()); f.write(M.tobytes())
        r = subprocess.run([gxx, "-O3", "-DNDEBUG", "-std=c++11", "-I", hits[0], str(ROOT / "oracle" / "eigen_probe.cc"), "-o", str(exe)], capture_output=True, text=True)
        rep["build_rc"] = r.returncode
        if r.returncode == 0:
            r = subprocess.run([str(exe), "--dump", str(fin), str(fout)], capture_output=True, text=True)
            rep["probe"] = r.stdout.strip()
            got = np.fromfile(fout, dtype=np.float64).reshape(3, k)
            v, vm, vmm = M[k + 2].copy(), M[k + 1].copy(), M[k].copy()
            scratch = (np.empty(k), np.empty(k), np.empty(k))
            oracle_lib.ref_scan_f64_eigen_gemv3(M, k, v, vm, vmm, 1, scratch)
            rep["bit_identical"] = [bool(np.array_equal(got[i].view(np.uint64), scratch[i].view(np.uint64))) for i in range(3)]
            rep["max_abs_diff"] = [float(np.abs(got[i] - scratch[i]).max()) for i in range(3)]
        else:
            rep["build_stderr"] = r.stderr[-600:]
    (out_dir / "eigen_pin.json").write_text(json.dumps(rep, indent=1) + "\n")
    print(json.dumps({k: rep[k] for k in rep if k not in ("searched_python_roots",)}, indent=1))


if __name__ == "__main__":
    main()
