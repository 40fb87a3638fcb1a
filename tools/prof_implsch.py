"""Profiling driver (not a test): runs IMPLSCH a few times on one configuration.  python tools/prof_implsch.py [sp|dp] [npoints] [generation 0|2|4] [A|B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
FLAGS = {"A": {}, "B": dict(llgcbz0=True, llnormagam=True)}[sys.argv[4] if len(sys.argv) > 4 else "A"]   # flag set
NANG = int(os.environ.get("ECWAM_PROF_NANG", "36"))      # 36 (the benchmark), 24, 12 or 48 directions
cfg = Config(nang=NANG, nfre=36, nfre_red=36, idelt=450, idelpro=450, **FLAGS)
case = H.make_point_case(4096, cfg, prec, spectra="mixed")
ctx = api.HipContext(case["tables"])
if len(sys.argv) > 3:
    ctx.set_implsch_generation(int(sys.argv[3]))
dev = ctx.device
wv, ff, intf = H.pack_device_inputs(case)
rep = (n + 4095) // 4096
fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)[:n].contiguous()
twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1)[:n].contiguous()
tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1)[:n].contiguous()
tin = torch.from_numpy(intf).to(dev).repeat(rep, 1)[:n].contiguous()
mij = torch.zeros(n, dtype=torch.int32, device=dev)
xl = torch.zeros_like(fl0)
for it in range(5):
    fl = fl0.clone()
    tff = tff0.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.implsch(0, n, fl, twv, tff, tin, mij, xl)
    e1.record()
    torch.cuda.synchronize()
    print("implsch ms", e0.elapsed_time(e1), flush=True)
