#!/usr/bin/env python3
"""diagnostics: time of the pieces of the sub-stepped advection (fused fast/slow pass, frequency-range copy, fast-wave pass)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from ecwam_amd import grid as G
from ecwam_amd.tables import Config
from ecwam_amd.wamintgr import Wamintgr

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dt = 450 if ng <= 320 else max(15, int(450 * 320 / ng) // 15 * 15)
cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=dt, idelpro=dt)
m = Wamintgr(cfg, G.build_grid(ng), "sp", ifrelfmax=5, delpro_lf=dt / 2.0)
m.init_synthetic()
assert m.build_weights() == 0
g = m.gd


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("full pass, one dt      :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(dt), 0, m.n, 1, 36)))
print("fused fast/slow pass   :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(dt), 0, m.n, 1, 36, ifrelfmax=5, delpro_lf=dt / 2.0)))
print("copy M=1..5            :", t(lambda: m.ctx.copy_freq_range(m.fl3, m.fl1, m.n, 1, 5)))
print("fast waves only (1..5) :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, dt / 2.0, 0, m.n, 1, 5, copy_rest=False)))
print("fast waves only (1..4) :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, dt / 2.0, 0, m.n, 1, 4, copy_rest=False)))
print("whole propag()         :", t(m.propag))
# round 3: the pieces of the compact-first order
print("fused pass + gout      :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(dt), 0, m.n, 1, 36, ifrelfmax=5, delpro_lf=dt / 2.0, gout=m.g1)))
print("sub-step compact->rows :", t(lambda: m.ctx.propags2_otf(m.g1, m.fl3, g, m.cgroup_ext, dt / 2.0, 0, m.n, 1, 5, copy_rest=False)))
print("extract M=1..8 compact :", t(lambda: m.ctx.copy_freq_range(m.fl1, m.g1, m.n, 1, 8)))
print("sub-step compact->comp :", t(lambda: m.ctx.propags2_otf(m.g1, m.g2, g, m.cgroup_ext, dt / 2.0, 0, m.n, 1, 5, copy_rest=True)))
print("full pass gin + gout   :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(dt), 0, m.n, 1, 36, ifrelfmax=5, delpro_lf=dt / 2.0, gin=m.g1, gout=m.g2)))
print("full pass gin          :", t(lambda: m.ctx.propags2_otf(m.fl1, m.fl3, g, m.cgroup_ext, float(dt), 0, m.n, 1, 36, ifrelfmax=5, delpro_lf=dt / 2.0, gin=m.g1)))
for mode in ("rows", "compact"):
    m.fast_mode = mode
    m.gfast_valid = False
    print(f"whole propag() {mode:8s}:", t(m.propag))
