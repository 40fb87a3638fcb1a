"""GPU tests at BASELINE.json's larger grids on ONE GPU (they fit 288 GB): O640 (config 4, 1 661 400 sea points) and O1280 (config 5,
6 599 640 sea points, single and double precision; 8.55e9 spectral bins per array: every index beyond 2**32).  One full WAMINTGR step
(PROPAGS2 with on-the-fly weights + IMPLSCH) on the device; the oracle on random samples: 400 points with their stencil neighbourhoods
for the advection, 400 points for IMPLSCH (no neighbour access), with the tolerances of test_gpu_parity.py."""
import numpy as np
import pytest

import harness as H
from ecwam_amd.tables import Config
from test_gpu_full_size import _sample_grid

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("ng,prec", [(640, "sp"), (1280, "sp"), (1280, "dp")])
def test_one_step_at_baseline_grid_matches_the_oracle_on_samples(ng, prec):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import grid as G
    from ecwam_amd.wamintgr import Wamintgr
    from oracle.oracle import Oracle

    need = {(640, "sp"): 40e9, (1280, "sp"): 140e9, (1280, "dp"): 250e9}[(ng, prec)]
    if torch.cuda.get_device_properties(0).total_memory < need:
        pytest.skip("not enough device memory")
    dt = 450 if ng <= 320 else max(15, int(450 * 320 / ng) // 15 * 15)      # bench.py's time step for the grid
    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=dt, idelpro=dt)
    g = G.build_grid(ng)
    n = g.nsea
    assert n == 4 * ng * (ng + 9) - 40
    m = Wamintgr(cfg, g, prec)
    m.init_synthetic()
    assert m.build_weights() == 0
    o = Oracle(cfg, prec)
    eps = np.finfo(m.npdt).eps
    rng = np.random.default_rng(ng)
    # ---- advection: the oracle on a sample with its neighbourhoods (the last points of the grid included: the largest offsets)
    sel = np.sort(np.concatenate([rng.choice(n - 64, 336, replace=False), np.arange(n - 64, n)]))
    sub, rows = _sample_grid(g, sel)
    tr = torch.from_numpy(rows).to(m.dev)
    fl_rows = m.fl1[tr].cpu().numpy()
    cg_rows = m.cgroup_ext[tr].cpu().numpy()
    cg_land = m.cgroup_ext[n:n + 1].cpu().numpy()
    flmax = float(m.fl1.max().item())
    m.propag()
    torch.cuda.synchronize()
    wref = o.ctu_weights(sub, np.concatenate([cg_rows, cg_land]), float(cfg.idelpro))
    f3ref = o.propags2(sub, np.concatenate([fl_rows, np.zeros_like(fl_rows[:1])]), wref)
    ts = torch.from_numpy(sel).to(m.dev)
    got = m.fl1[ts].cpu().numpy()
    assert np.max(np.abs(got.astype(float) - f3ref[: sel.size].astype(float))) < 16 * eps * flmax
    # (reductions only: an element-wise isfinite of the 68 GB double-precision array would need as much again for its temporaries)
    assert float(m.fl1[:n].min().item()) >= 0.0 and np.isfinite(float(m.fl1[:n].sum(dtype=torch.float64).item()))
    # ---- IMPLSCH: the oracle on a sample of the advected points
    wv = m.wvprpt[ts].cpu().numpy()
    ffh = m.ff[ts].cpu().numpy()
    case = dict(cfg=cfg, prec=prec, tables=m.t, n=sel.size, FL1=got.copy(),
                props=dict(WAVNUM=wv[:, 0], CGROUP=wv[:, 1], CINV=wv[:, 2], XK2CG=wv[:, 3], STOKFAC=wv[:, 4]),
                FF=ffh[:, :14], ENV=ffh[:, 14:16], INTF=m.intf[ts].cpu().numpy()[:, :15])
    ref = H.oracle_implsch(case, o)
    m.implsch()
    torch.cuda.synchronize()
    out = dict(FL1=m.fl1[ts].cpu().numpy(), XLLWS=m.xllws[ts].cpu().numpy(), MIJ=m.mij[ts].cpu().numpy(), FF=m.ff[ts].cpu().numpy()[:, :14],
               INTF=m.intf[ts].cpu().numpy()[:, :15])
    st = H.compare_implsch(ref, out, m.t)
    ns = sel.size
    if prec == "dp":
        assert st["mij_flips"] == 0 and st["xllws_bins_diff"] == 0, st
        assert st["fl1_max_rel_peak_all"] < 1e-10 and st["swh_max_rel"] < 1e-12, st
        assert st["ff_max_rel_all"] < 1e-10 and st["intf_max_rel_all"] < 1e-8, st
    else:
        H.assert_sp_gates(st, ns)      # (the single-precision gates for time steps of at most 450 s, harness.SP_GATES["short"]: these grids run 225 s and 105 s)
    assert np.isfinite(float(m.fl1[:n].sum(dtype=torch.float64).item())) and float(m.fl1[:n].min().item()) >= 0.0
    assert int(m.mij.min().item()) >= 1 and int(m.mij.max().item()) <= cfg.nfre
    m.ctx.close()
    del m
    torch.cuda.empty_cache()
