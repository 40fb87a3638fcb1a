"""Diagnostics (not a test): one step of the one-kernel path against the two kernels on a small grid with land, for a list of
(directions, NFRE_RED, precision, IFRELFMAX, LSUBGRID) cases; prints how many bins / points differ and where (per direction, per frequency).
What showed that the double precision one-kernel builds at 12 / 24 directions are miscompiled at -O3 (profiles/r06_fused_step_experiments.txt):
with every direction count enabled in implsch4a.hip, run it on the product library and on ECWAM_HIP_LIB=<...>/libecwam_hip_advO1.so.
python tools/dbg_fused.py"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ecwam_amd import grid as G
from ecwam_amd.tables import Config
from ecwam_amd.wamintgr import Wamintgr
def run(nang, nred, prec, lfm=0, subgrid=False):
    cfg = Config(nang=nang, nfre=36, nfre_red=nred, idelt=450, idelpro=450)
    g = G.build_grid(17, mask="continents")
    kw = dict(ifrelfmax=lfm, delpro_lf=225.0) if lfm else {}
    ms = []
    for _ in range(2):
        m = Wamintgr(cfg, g, prec, **kw); m.init_synthetic(seed=29); ms.append(m)
    two, one = ms
    if subgrid:
        from ecwam_amd import synthetic as syn
        obs = syn.obstructions(g, 36, seed=5); obs[:, :, nred:] = 1.0
        for m in ms: m.set_obstructions(obs)
    assert two.build_weights() == 0 and one.build_weights() == 0
    # no source terms: compare advection only through LLSOURCE... use IMPLSCH anyway
    two.step(); one.step(fused=True); torch.cuda.synchronize()
    a = two.fl1[:two.n].cpu().numpy(); b = one.fl1[:one.n].cpu().numpy()
    d = a != b
    print(nang, nred, prec, lfm, subgrid, "differ", d.sum(), "of", d.size, "points", d.any(axis=(1,2)).sum(), "of", two.n)
    if d.any():
        pts = np.flatnonzero(d.any(axis=(1,2)))
        print(" first points", pts[:20], " pts mod PP", np.bincount(pts % 5, minlength=5))
        print(" per direction", d.sum(axis=(0,2)))
        print(" per frequency", d.sum(axis=(0,1)))
        rel = np.abs(a-b)/np.maximum(np.abs(a),1e-300)
        print(" max rel", rel.max(), " mij equal", bool(torch.equal(two.mij, one.mij)))
    for m in ms: m.ctx.close()
run(12, 25, "dp"); run(12, 25, "sp"); run(12, 36, "dp"); run(24, 29, "dp", 5, True); run(24, 29, "dp", 5, False); run(24, 29, "dp", 0, True)
