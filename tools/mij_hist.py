#!/usr/bin/env python3
"""diagnostics: distribution of the cut-off index MIJ (and of its maximum over the sea points a wavefront of k_implsch4 carries) on the
benchmark's synthetic state, after s steps.  usage: python tools/mij_hist.py [grid] [steps] [points per wave]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecwam_amd import grid as G
from ecwam_amd.tables import Config
from ecwam_amd.wamintgr import Wamintgr

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 320
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pp = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m = Wamintgr(Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450), G.build_grid(ng), "sp")
m.init_synthetic()
assert m.build_weights() == 0
for s in range(steps):
    m.step()
    mij = m.mij[: m.n].to(torch.int64).cpu()
    n3 = (m.n // pp) * pp
    mx = mij[:n3].view(-1, pp).max(dim=1).values
    h = torch.bincount(mij, minlength=37)[1:]
    hx = torch.bincount(mx, minlength=37)[1:]
    print(f"step {s + 1}: MIJ mean {mij.float().mean():.2f}; wave maximum mean {mx.float().mean():.2f}")
    print("  MIJ      :", " ".join(f"{int(v)}" for v in h))
    print("  wave max :", " ".join(f"{int(v)}" for v in hx))
    for cut in (19, 23, 27, 31):
        print(f"  waves with max MIJ <= {cut}: {float((mx <= cut).float().mean()):.3f}")
