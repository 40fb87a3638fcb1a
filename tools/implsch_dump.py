"""Diagnostic (not a test): one IMPLSCH call on a seeded case, outputs saved for a bit-by-bit comparison of two builds of the library.
ECWAM_HIP_LIB=<library> python tools/implsch_dump.py [sp|dp] [npoints] out.npz [A|B]   /   python tools/implsch_dump.py --compare a.npz b.npz"""
import os
import sys

import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        x, y = a[k], b[k]
        same = np.array_equal(x.view(np.uint8), y.view(np.uint8))
        d = np.abs(x.astype(np.float64) - y.astype(np.float64))
        nz = int(np.count_nonzero(x != y))
        print(f"{k}: {'identical' if same else 'DIFFERENT'}  elements differing {nz} of {x.size}  max abs {d.max():.3e}  max rel to peak {d.max() / max(np.abs(x).max(), 1e-300):.3e}")
        bad += 0 if same else 1
    sys.exit(1 if bad else 0)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402

prec, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
FLAGS = {"A": {}, "B": dict(llgcbz0=True, llnormagam=True)}[sys.argv[4] if len(sys.argv) > 4 else "A"]
cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450, **FLAGS)
case = H.make_point_case(n, cfg, prec, spectra="mixed")
ctx = api.HipContext(case["tables"])
dev = ctx.device
wv, ff, intf = H.pack_device_inputs(case)
fl = torch.from_numpy(case["FL1"]).to(dev)
twv, tff, tin = torch.from_numpy(wv).to(dev), torch.from_numpy(ff).to(dev), torch.from_numpy(intf).to(dev)
mij = torch.zeros(n, dtype=torch.int32, device=dev)
xl = torch.zeros_like(fl)
ctx.implsch(0, n, fl, twv, tff, tin, mij, xl)
torch.cuda.synchronize()
np.savez(out, fl=fl.cpu().numpy(), ff=tff.cpu().numpy(), intf=tin.cpu().numpy(), mij=mij.cpu().numpy(), xllws=xl.cpu().numpy())
print("saved", out)
