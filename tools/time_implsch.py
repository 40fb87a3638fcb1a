"""Timing diagnostics (not a test): IMPLSCH kernel time with phases ablated through ECWAM_HIP_DEBUG_SKIP.
Each mask runs in a child process because the mask is read when the context is created."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import harness as H
from ecwam_amd import api
from ecwam_amd.tables import Config
prec = sys.argv[2]; n = int(sys.argv[3])
cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
case = H.make_point_case(4096, cfg, prec, spectra="mixed")
ctx = api.HipContext(case["tables"])
dev = ctx.device
wv, ff, intf = H.pack_device_inputs(case)
rep = n // 4096
fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)
twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1); tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1); tin = torch.from_numpy(intf).to(dev).repeat(rep, 1)
mij = torch.zeros(n, dtype=torch.int32, device=dev); xl = torch.zeros_like(fl0)
ts = []
for it in range(4):
    fl = fl0.clone(); tff = tff0.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.implsch(0, n, fl, twv, tff, tin, mij, xl); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{min(ts[1:]):.3f}")
'''


def run(mask, prec, n):
    env = dict(os.environ, ECWAM_HIP_DEBUG_SKIP=str(mask))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, prec, str(n)], env=env, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.returncode == 0 else "ERR " + r.stderr[-300:]


if __name__ == "__main__":
    n = 131072
    names = {0: "full", 1: "-sinput", 2: "-stresso", 4: "-sdissip", 8: "-snonlin", 16: "-taut_z0", 32: "-update", 127: "skeleton(load/store+means)"}
    for prec in sys.argv[1:] or ["sp"]:
        for mask, nm in names.items():
            print(prec, nm, run(mask, prec, n), "ms for", n, "points", flush=True)
