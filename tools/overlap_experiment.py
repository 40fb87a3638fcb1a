"""Diagnostics: WAMINTGR at O320 with the advection of block c+1 running beside the source terms of block c on a second stream.
usage: python3 tools/overlap_experiment.py sp|dp [nchunks ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecwam_amd import grid as G  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
from ecwam_amd.wamintgr import Wamintgr  # noqa: E402

cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
g = G.build_grid(320)
prec = sys.argv[1] if len(sys.argv) > 1 else "sp"
w = Wamintgr(cfg, g, prec)
w.init_synthetic()
w.build_weights()
n = w.n
sP, sI = torch.cuda.Stream(), torch.cuda.Stream()


def step_serial():
    w.step()


def step_pipelined(nch):
    c = w.cfg
    b = [n * i // nch for i in range(nch + 1)]
    cur = torch.cuda.current_stream()
    sP.wait_stream(cur)
    sI.wait_stream(cur)
    evs = []
    with torch.cuda.stream(sP):
        for i in range(nch):
            w.ctx.propags2_otf(w.fl1, w.fl3, w.gd, w.cgroup_ext, float(c.idelpro), b[i], b[i + 1], 1, c.nfre_red, copy_rest=True)
            e = torch.cuda.Event()
            e.record(sP)
            evs.append(e)
    with torch.cuda.stream(sI):
        for i in range(nch):
            sI.wait_event(evs[i])
            w.ctx.implsch(b[i], b[i + 1], w.fl3, w.wvprpt, w.ff, w.intf, w.mij, w.xllws)
    cur.wait_stream(sI)
    cur.wait_stream(sP)
    w.fl1, w.fl3 = w.fl3, w.fl1


def timeit(f, *a, reps=10):
    for _ in range(2):
        f(*a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print("serial ms", timeit(step_serial), flush=True)
for nch in [int(a) for a in sys.argv[2:]] or [2, 4, 8, 16, 32]:
    print("pipelined", nch, "ms", timeit(step_pipelined, nch), flush=True)
print("serial ms", timeit(step_serial), flush=True)
print("norm", w.swh_norm())
