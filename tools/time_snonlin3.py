"""Time SNONLIN in the three-points-per-wavefront layout (ecwam_hip_snonlin3) against the DIA share of k_implsch2.

    python tools/time_snonlin3.py [npoints]

Prints the kernel time with the DIA (mode 0) and with load / store only (mode 1); their difference is what the DIA costs in this
layout.  k_implsch2's DIA costs 0.96 ms per 131072 points (profiles/r01_implsch_phase_instruction_counts.txt: 4.92 ms with,
3.96 ms without)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    cfg = Config(nang=36, nfre=36, nfre_red=36)
    base = H.make_point_case(4096, cfg, "sp", spectra="mixed", seed=3)
    ctx = api.HipContext(base["tables"])
    rep = (n + 4095) // 4096
    fl = torch.from_numpy(np.tile(base["FL1"], (rep, 1, 1))[:n].copy()).to(ctx.device)
    depth = torch.from_numpy(np.tile(base["ENV"][:, 1], rep)[:n].astype(np.float32)).to(ctx.device)
    ak = torch.from_numpy(np.tile(base["props"]["WAVNUM"][:, 10], rep)[:n].astype(np.float32)).to(ctx.device)
    out = {}
    for mode in (1, 0):
        for _ in range(3):
            ctx.snonlin3(fl, depth, ak, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ctx.snonlin3(fl, depth, ak, mode)
        e1.record()
        torch.cuda.synchronize()
        out[mode] = e0.elapsed_time(e1) / reps
    print(f"points {n}: load/store only {out[1]:.3f} ms, with the DIA {out[0]:.3f} ms -> DIA {out[0] - out[1]:.3f} ms "
          f"({(out[0] - out[1]) * 131072 / n:.3f} ms per 131072 points; k_implsch2: 0.96 ms)")
    # SINPUT_ARD, second SINFLX call (two gust states): k_implsch2 spends 1.28 ms per 131072 points on its three state evaluations
    # (4.92 ms with, 3.64 ms without SINPUT), i.e. ~0.85 ms on the two of this call
    pr, ff = base["props"], base["FF"]
    wv = np.stack([pr[k] for k in ("WAVNUM", "CGROUP", "CINV", "XK2CG", "STOKFAC")], 1).astype(np.float32)
    pt = np.zeros((4096, 12), np.float32)
    pt[:, 0], pt[:, 1] = ff[:, 7], ff[:, 10]
    pt[:, 2] = np.maximum(ff[:, 0], 1.0) * float(base["tables"].ROWATERM1)
    pt[:, 3], pt[:, 4], pt[:, 5], pt[:, 6] = 0.1, 0.05, 0.5, 0.0006
    pt[:, 7], pt[:, 8] = np.sin(ff[:, 1]), np.cos(ff[:, 1])
    twv = torch.from_numpy(np.tile(wv, (rep, 1, 1))[:n].copy()).to(ctx.device)
    tpt = torch.from_numpy(np.tile(pt, (rep, 1))[:n].copy()).to(ctx.device)
    for mode in (1, 0):
        for _ in range(3):
            ctx.sinput3(fl, twv, tpt, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.sinput3(fl, twv, tpt, mode)
        e1.record()
        torch.cuda.synchronize()
        out[mode] = e0.elapsed_time(e1) / 20
    print(f"points {n}: load/store only {out[1]:.3f} ms, with SINPUT_ARD (2 gust states) {out[0]:.3f} ms -> {out[0] - out[1]:.3f} ms "
          f"({(out[0] - out[1]) * 131072 / n:.3f} ms per 131072 points; k_implsch2: ~0.85 ms for this call)")

    # SDISSIP_ARD: k_implsch2 spends 0.63 ms per 131072 points on it (4.92 ms with, 4.29 ms without)
    for mode in (1, 0):
        for _ in range(3):
            ctx.sdissip3(fl, twv, tpt, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.sdissip3(fl, twv, tpt, mode)
        e1.record()
        torch.cuda.synchronize()
        out[mode] = e0.elapsed_time(e1) / 20
    print(f"points {n}: load/store only {out[1]:.3f} ms, with SDISSIP_ARD {out[0]:.3f} ms -> {out[0] - out[1]:.3f} ms "
          f"({(out[0] - out[1]) * 131072 / n:.3f} ms per 131072 points; k_implsch2: 0.63 ms)")


if __name__ == "__main__":
    main()
