"""Time per step of the drop-in host path (Fortran WAMINTGR_HIP -> iso_c_binding -> C ABI) next to the Python host driving the same C
ABI, same grid, same spectra (diagnostics, not a test).  python tools/time_fortran_host.py [grid=320] [nstep=12] [prec=sp]"""
import os
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fortran import _write_case  # noqa: E402
from ecwam_amd import build, grid as G  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
from ecwam_amd.wamintgr import Wamintgr  # noqa: E402

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 320
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prec = sys.argv[3] if len(sys.argv) > 3 else "sp"
cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
g = G.build_grid(ng)
m = Wamintgr(cfg, g, prec)
m.init_synthetic()
assert m.build_weights() == 0
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
case, out = os.path.join(d, "case.bin"), os.path.join(d, "out.bin")
_write_case(case, m, cfg, g, 32, nstep)
print(f"case file {os.path.getsize(case) / 1e9:.2f} GB, {g.nsea} points", flush=True)
for _ in range(2):
    m.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(nstep - 2):
    m.step()
torch.cuda.synchronize()
tp = (time.perf_counter() - t0) / (nstep - 2) * 1e3
print(f"python host: {tp:.3f} ms per step", flush=True)
del m
torch.cuda.empty_cache()
exe = build.fortran_exe(prec)
r = subprocess.run([exe, case, out, "time"], capture_output=True, text=True, timeout=1200)
print(r.stdout.strip(), r.stderr.strip()[-500:], flush=True)
os.remove(case)
if os.path.exists(out):
    os.remove(out)
