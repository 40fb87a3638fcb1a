"""Diagnostics (not a test): IMPLSCH kernel time against the number of resident waves per CU, lowered by padding the dynamic LDS
request (ECWAM_HIP_IMPLSCH_PADLDS).  Sizes the "issue time vs exposed latency" of one wave (plan for the next kernel)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from time_implsch import CHILD, ROOT  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
base = 3 * 2 * 36 * 36 * (4 if prec == "sp" else 8) + 3 * 46 * (4 if prec == "sp" else 8)     # bytes of a 3-wave block
for blocks in (5, 4, 3, 2, 1):
    pad = max(0, 160 * 1024 // blocks - base - 256) if blocks < 5 else 0
    env = dict(os.environ, ECWAM_HIP_IMPLSCH_PADLDS=str(pad))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, prec, str(n)], env=env, capture_output=True, text=True)
    out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 else "ERR " + r.stderr[-300:]
    print(f"blocks/CU {blocks}  waves/CU {3 * blocks:2d}  pad {pad:6d} B  ms {out}", flush=True)
