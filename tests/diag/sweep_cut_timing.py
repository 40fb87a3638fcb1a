"""diagnostics: what ending the sweep at the cut-off frequency is worth -- IMPLSCH per launch with the cut (LWFLUX = F) and without it
(LWFLUX = T runs every interaction) on two states: the benchmark's synthetic one and long swell under strong winds.
python tests/diag/sweep_cut_timing.py [npoints]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api, synthetic as syn  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for state in ("benchmark", "swell under strong winds"):
    for lw in (False, True):
        cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450, lwflux=lw)
        case = H.make_point_case(n, cfg, "sp", seed=12345)
        if state != "benchmark":
            rng = np.random.default_rng(7)
            case["FL1"] = syn.jonswap_spectra(case["tables"].FR, case["tables"].TH, rng.uniform(0.045, 0.09, n), rng.uniform(0, 2 * np.pi, n), np.float32)
            case["params"]["WSWAVE"] = rng.uniform(12.0, 35.0, n)
            case["params"]["CICOVER"] = np.zeros(n)
            case["FF"] = syn.forcing(case["params"], slice(0, n), case["tables"], np.float32)
        ctx = api.HipContext(case["tables"])
        wv, ff, intf = H.pack_device_inputs(case)
        dev = ctx.device
        fl0 = torch.from_numpy(case["FL1"]).to(dev)
        twv, tff0, tintf0 = (torch.from_numpy(a).to(dev) for a in (wv, ff, intf))
        mij = torch.zeros(n, dtype=torch.int32, device=dev)
        xl = torch.zeros_like(fl0)
        ts = []
        for it in range(5):
            fl, tff, tintf = fl0.clone(), tff0.clone(), tintf0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.implsch(0, n, fl, twv, tff, tintf, mij, xl)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        m = mij.cpu().numpy()
        mx = m[: n // 3 * 3].reshape(-1, 3).max(1)
        print(f"{state:26s} LWFLUX={'T' if lw else 'F'}: {min(ts[1:]):.3f} ms / {n} points; MIJ mean {m.mean():.1f}, wave maximum mean {mx.mean():.1f}")
        ctx.close()
