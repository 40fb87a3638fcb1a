"""Experiment: software pipeline of one WAMINTGR step over latitude bands on two HIP streams.

PROPAGS2 is bound by the latency of its gathers, IMPLSCH by vector-ALU issue: run side by side they
could share a CU.  PROPAGS2 of band j (step t+1) reads rows of bands j-1..j+1 of the spectra IMPLSCH(t) updated in place and
writes the other buffer, so it may start as soon as IMPLSCH(t) has finished band j+1; IMPLSCH(t+1) of band j only needs
PROPAGS2(t+1) of band j.  This script times that schedule against the serial one and checks that the spectra are identical.

    python tools/pipeline_experiment.py [--grid 320] [--bands 2,4,8,16] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ecwam_amd import grid as G  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
from ecwam_amd.wamintgr import Wamintgr  # noqa: E402


def run_serial(m, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run_pipelined(m, steps, nb, prio):
    c, g, n = m.cfg, m.gd, m.n
    edges = [n * j // nb for j in range(nb + 1)]
    s_src = torch.cuda.current_stream()
    s_adv = torch.cuda.Stream(priority=-1 if prio else 0)
    src_done = [None] * nb            # IMPLSCH(t) finished band j
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        adv_done = []
        for j in range(nb):
            k0, k1 = edges[j], edges[j + 1]
            with torch.cuda.stream(s_adv):
                for jj in (j - 1, j, j + 1):          # neighbours' rows updated by the previous source step
                    if 0 <= jj < nb and src_done[jj] is not None:
                        s_adv.wait_event(src_done[jj])
                m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(c.idelpro), k0, k1, 1, c.nfre_red, copy_rest=True)
                e = torch.cuda.Event()
                e.record(s_adv)
                adv_done.append(e)
        new_done = []
        for j in range(nb):
            k0, k1 = edges[j], edges[j + 1]
            s_src.wait_event(adv_done[j])
            m.ctx.implsch(k0, k1, m.fl3, m.wvprpt, m.ff, m.intf, m.mij, m.xllws)
            e = torch.cuda.Event()
            e.record(s_src)
            new_done.append(e)
        src_done = new_done
        m.fl1, m.fl3 = m.fl3, m.fl1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=320)
    ap.add_argument("--bands", default="2,4,8,16,32")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    g = G.build_grid(a.grid, mask="aqua")

    def fresh():
        m = Wamintgr(cfg, g, "sp")
        m.init_synthetic()
        assert m.build_weights() == 0
        return m

    m = fresh()
    run_serial(m, 2)
    print(f"serial: {run_serial(m, a.steps):.3f} ms per step", flush=True)
    ref = fresh()
    for _ in range(3):
        ref.step()
    torch.cuda.synchronize()
    for nb in [int(x) for x in a.bands.split(",")]:
        for prio in (0, 1):
            p = fresh()
            run_pipelined(p, 3, nb, prio)
            same = bool(torch.equal(p.fl1[: p.n], ref.fl1[: ref.n]))
            ms = run_pipelined(p, a.steps, nb, prio)
            print(f"bands {nb:3d} advect-priority {prio}: {ms:.3f} ms per step, spectra identical to serial after 3 steps: {same}", flush=True)
            del p
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
