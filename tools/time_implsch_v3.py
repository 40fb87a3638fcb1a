"""IMPLSCH kernel time, k_implsch2 against the three-points-per-wavefront kernel (ECWAM_HIP_IMPLSCH_V3=1), same inputs, and the
largest difference between their outputs.  python tools/time_implsch_v3.py [npoints]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
case = H.make_point_case(4096, cfg, "sp", spectra="mixed")
ctx = api.HipContext(case["tables"])
dev = ctx.device
wv, ff, intf = H.pack_device_inputs(case)
rep = (n + 4095) // 4096
fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)[:n].contiguous()
twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1)[:n].contiguous()
tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1)[:n].contiguous()
tin0 = torch.from_numpy(intf).to(dev).repeat(rep, 1)[:n].contiguous()
res = {}
for v3 in (0, 1):  # 0: k_implsch2, 1 (the default): k_implsch3
    os.environ["ECWAM_HIP_IMPLSCH_V3"] = str(v3)
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
    res[v3] = (min(ts[1:]), fl, tff, tin, mij, xl)
    print(f"{'k_implsch3' if v3 else 'k_implsch2'}: {min(ts[1:]):.3f} ms for {n} points", flush=True)
a, b = res[0], res[1]
pk = a[1].abs().amax(dim=(1, 2), keepdim=True)
print("max |dF| / peak:", float(((a[1] - b[1]).abs() / pk).max()), " MIJ differ:", int((a[4] != b[4]).sum()), " XLLWS bins differ:",
      int((a[5] != b[5]).sum()), " max |dFF|:", float((a[2] - b[2]).abs().max()), " max |dINTF|:", float((a[3] - b[3]).abs().max()))
print(f"speed-up {a[0] / b[0]:.2f}x")
