"""Diagnostic (not a test): test_implsch_in_blocks_is_bit_identical with a report of WHERE the block-wise result differs from the whole call.
python tests/diag/blocks_probe.py [sp|dp] [streams 1|2]"""
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

prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = Config(nang=36, nfre=36, nfre_red=36)
n = 3001
case = H.make_point_case(n, cfg, prec, spectra="mixed", seed=31)
ctx = api.HipContext(case["tables"])
whole = H.gpu_implsch(case, ctx)
whole2 = H.gpu_implsch(case, ctx)
print("whole call twice identical:", all(np.array_equal(whole[k], whole2[k]) for k in ("FL1", "XLLWS", "MIJ", "FF", "INTF")))
ctx.close()
ctx = api.HipContext(case["tables"])
dev = ctx.device
wv, ff, intf = H.pack_device_inputs(case)
fl1 = torch.from_numpy(case["FL1"].copy()).to(dev)
twv, tff, tintf = (torch.from_numpy(a).to(dev) for a in (wv, ff, intf))
mij = torch.zeros(n, dtype=torch.int32, device=dev)
xllws = torch.zeros_like(fl1)
bounds = [0, 7, 64, 65, 1000, 1001, 2048, n]
streams = [torch.cuda.Stream() for _ in range(nstreams)]
torch.cuda.synchronize()
for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
    with torch.cuda.stream(streams[i % nstreams]):
        ctx.implsch(a, b, fl1, twv, tff, tintf, mij, xllws)
torch.cuda.synchronize()
got = dict(FL1=fl1.cpu().numpy(), XLLWS=xllws.cpu().numpy(), MIJ=mij.cpu().numpy(), FF=tff.cpu().numpy()[:, :14], INTF=tintf.cpu().numpy()[:, :15])
ctx.close()
for k in ("FL1", "XLLWS", "MIJ", "FF", "INTF"):
    a, b = whole[k], got[k]
    d = (a != b).reshape(n, -1).any(axis=1)
    idx = np.flatnonzero(d)
    print(k, "points differing:", idx.size, idx[:20], "..." if idx.size > 20 else "")
    if idx.size and k == "FL1":
        i = idx[0]
        e = np.abs(a[i].astype(float) - b[i].astype(float))
        print("   first differing point", i, "max abs diff", e.max(), "peak", np.abs(a[i]).max(), "bins differing", int((a[i] != b[i]).sum()), "rows (M) differing", np.flatnonzero((a[i] != b[i]).any(axis=0))[:40])
