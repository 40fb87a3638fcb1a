"""IMPLSCH kernel generations side by side (diagnostics, not a test): time per launch and parity statistics against the oracle.
python tests/diag/implsch_gens.py [npoints] [prec,...] [nang,...] [A|B|N|G|J|E: flag set A, B (LLGCBZ0 + LLNORMAGAM), LLNORMAGAM only, LLGCBZ0 only, IPHYS = 0, ISNONLIN = 1;
the RARE builds: R2 = ISNONLIN 2, RI = LCIWA1 + LCIWA2, RU = ICODE 1, RW = LWVFLX_SNL = F, RB = flag set B + ISNONLIN 1, RJ = IPHYS 0 + ISNONLIN 1]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["sp", "dp"]
nangs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [36, 24, 12]
NRED = {36: 36, 24: 29, 12: 25}
FLAGS = {"A": {}, "B": dict(llgcbz0=True, llnormagam=True), "N": dict(llnormagam=True), "G": dict(llgcbz0=True), "J": dict(iphys=0),
         "E": dict(isnonlin=1), "R2": dict(isnonlin=2),
         "RI": dict(lciwa1=True, lciwa2=True, lmaskice=False),
         "RU": dict(icode=1), "RW": dict(lwvflx_snl=False), "RB": dict(llgcbz0=True, llnormagam=True, isnonlin=1), "RJ": dict(iphys=0, isnonlin=1)}[sys.argv[4] if len(sys.argv) > 4 else "A"]
for nang in nangs:
    for prec in precs:
        cfg = Config(nang=nang, nfre=36, nfre_red=NRED[nang], idelt=450, idelpro=450, **FLAGS)
        nc = 1537
        case = H.make_point_case(nc, cfg, prec, spectra="mixed", seed=777)
        ref = H.oracle_implsch(case, Oracle(cfg, prec))
        ctx = api.HipContext(case["tables"])
        dev = ctx.device
        wv, ff, intf = H.pack_device_inputs(case)
        rep = (n + nc - 1) // nc
        fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)[:n].contiguous()
        twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1)[:n].contiguous()
        tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1)[:n].contiguous()
        tin0 = torch.from_numpy(intf).to(dev).repeat(rep, 1)[:n].contiguous()
        for gen in (2, 4):
            ctx.set_implsch_generation(gen)
            got = H.gpu_implsch(case, ctx)
            st = H.compare_implsch(ref, got, case["tables"])
            ts = []
            for it in range(4):
                fl, tff, tin = fl0.clone(), tff0.clone(), tin0.clone()
                mij = torch.zeros(n, dtype=torch.int32, device=dev)
                xl = torch.zeros_like(fl0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.implsch(0, n, fl, twv, tff, tin, mij, xl)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            keys = ["mij_flips", "xllws_pts_diff", "fl1_max_rel_peak_clean", "fl1_max_rel_peak_all", "fl1_frac_bins_gt_1e-5", "swh_max_rel",
                    "ff_max_rel_clean", "intf_max_rel_clean", "intf_worst_group"]
            print(f"nang={nang} {prec} gen={gen}: {min(ts[1:]):.3f} ms / {n} points  " + json.dumps({k: st[k] for k in keys}), flush=True)
        ctx.close()
