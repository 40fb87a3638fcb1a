"""Diagnostics (not a test): k_implsch4 time against the number of resident waves per CU, lowered by padding the dynamic LDS request
(ECWAM_HIP_IMPLSCH_PADLDS, implsch_v4_launch.h).  The product runs 8 waves per CU in single precision (20 448 B each), 4 in double.
python tools/occupancy_sweep4.py [sp|dp] [npoints]   ->  profiles/r05_implsch4_occupancy_sweep.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
w = 4 if prec == "sp" else 8
base = ((36 + 4) * 3 * 36 + 3 * 36 * 6 + 3 * 48) * w      # v4_lds_bytes<T, 36, 3>(), NSC = 48
for waves in ((8, 7, 6, 5, 4, 3, 2) if prec == "sp" else (4, 3, 2)):
    pad = 0 if waves == (8 if prec == "sp" else 4) else max(0, 160 * 1024 // waves - base - 64)
    env = dict(os.environ, ECWAM_HIP_IMPLSCH_PADLDS=str(pad))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_implsch.py"), prec, str(n), "4"], env=env, capture_output=True, text=True)
    ms = [float(x.split()[-1]) for x in r.stdout.splitlines() if x.startswith("implsch ms")]
    out = f"{min(ms):.3f} (min of {len(ms)})" if r.returncode == 0 and ms else "ERR " + r.stderr[-300:]
    print(f"waves/CU {waves}  waves/SIMD {waves / 4:.2f}  pad {pad:6d} B  ms {out}", flush=True)
