"""Profiling driver: IMPLSCH time for flag set A vs B (sp, 36x36, 131072 points)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
for name, kw in (("A", {}), ("A+NORMAGAM", dict(llnormagam=True)), ("B", dict(llgcbz0=True, llnormagam=True))):
    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450, **kw)
    case = H.make_point_case(4096, cfg, prec, spectra="mixed")
    ctx = api.HipContext(case["tables"])
    dev = ctx.device
    wv, ff, intf = H.pack_device_inputs(case)
    n, rep = 131072, 32
    fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)
    twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1)
    tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1)
    tin = torch.from_numpy(intf).to(dev).repeat(rep, 1)
    mij = torch.zeros(n, dtype=torch.int32, device=dev)
    xl = torch.zeros_like(fl0)
    best = 1e9
    for it in range(3):
        fl, tff = fl0.clone(), tff0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.implsch(0, n, fl, twv, tff, tin, mij, xl); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"flag set {name}: implsch {best:.3f} ms per {n} points", flush=True)
    ctx.close()
