"""GPU parity of the refraction paths (IREFRA = 1 depth, 2 currents, 3 depth + currents): GRADI + PROPDOT, the CTUWDRV
checks with the LLCFLCUROFF second call, and PROPAGS2 with every weight rebuilt on the fly, against the oracle's
restatement of the reference's stored-weight scheme (gradi.F90, propdot.F90, ctuwdrv.F90, ctuw.F90, propags2.F90:124-192).

Tolerances: the device forms each weight with the reference's operations in the reference's order (contraction off), so
differences come from the division / reciprocal roundings only: theta-dot terms within 16 eps of their scale, F3 within
32 eps of the spectrum's maximum; CFL flags and CURMASK identical.
"""
import numpy as np
import pytest

import harness as H
from ecwam_amd.tables import Config, Tables

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def api():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import api as _api

    return _api


def _case(prec, irefra, n_oct=20, nang=24, nred=29, delpro=600, cur_amp=0.8, smooth_depth=True):
    from ecwam_amd import grid as G, synthetic as syn

    g = G.build_grid(n_oct, mask="continents")
    cfg = Config(nang=nang, nfre=36, nfre_red=nred, idelpro=delpro, idelt=delpro, irefra=irefra)
    dt = H.np_dtype(prec)
    t = Tables(cfg, dt)
    n = g.nsea
    rng = np.random.default_rng(5)
    lat = np.deg2rad(g.lat_deg)
    lon = np.deg2rad(g.ixlg * g.zdello[g.kxlt])
    if smooth_depth:   # shelf-like bathymetry: 20 m .. 900 m, smooth on the grid scale so that the depth refraction respects the CFL limit
        depth = 460.0 + 440.0 * np.sin(2 * lon + 1.0) * np.cos(3 * lat)
    else:
        depth = np.where(rng.uniform(0, 1, n) < 0.3, 10 ** rng.uniform(0.5, 3, n), 998.999)
    depth = np.minimum(depth, 998.999)
    props = syn.depth_props(depth, t, dt)
    land = syn.depth_props(np.array([998.999]), t, dt)

    def ext(a, lv):
        e = np.zeros((n + 1,) + a.shape[1:], dt)
        e[:n] = a
        e[n] = lv
        return e

    u, v = syn.currents(g, amp=cur_amp)
    if irefra < 2:
        u, v = np.zeros(n), np.zeros(n)
    c = dict(g=g, cfg=cfg, t=t, n=n, cg=ext(props["CGROUP"], land["CGROUP"][0]), om=ext(props["OMOSNH2KD"], land["OMOSNH2KD"][0]),
             wn=ext(props["WAVNUM"], land["WAVNUM"][0]), dep=ext(depth.astype(dt), 998.999), u=ext(u.astype(dt), 0.0), v=ext(v.astype(dt), 0.0))
    f1 = np.zeros((n + 1, cfg.nang, cfg.nfre), dt)
    f1[:n] = rng.uniform(0, 1, (n, cfg.nang, cfg.nfre)) ** 4
    c["f1"] = f1
    return c


def _oracle_run(c, prec, llcflcuroff=True):
    from oracle.oracle import Oracle

    o = Oracle(c["cfg"], prec)
    ir = c["cfg"].irefra
    dot = o.propdot(c["g"], ir, c["dep"], c["u"], c["v"], c["wn"], c["cg"], c["om"])
    w = o.ctu_weights_gen(c["g"], ir, c["cg"], c["om"], c["u"], c["v"], dot, float(c["cfg"].idelpro), llcflcuroff=llcflcuroff)
    f3 = o.propags2_gen(c["g"], c["f1"], w) if ir >= 2 else o.propags2(c["g"], c["f1"], w)
    return dot, w, f3


def _device_run(api, c, llcflcuroff=True):
    ctx = api.HipContext(c["t"])
    dev, cfg, n = ctx.device, c["cfg"], c["n"]
    gd = api.grid_to_device(c["g"], ctx.dtype, dev)
    td = {k: torch.from_numpy(c[k]).to(dev) for k in ("cg", "om", "wn", "dep", "u", "v", "f1")}
    refr = torch.zeros((n, 2 * cfg.nang + 5), dtype=ctx.dtype, device=dev)
    fail = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.propdot(gd, td["dep"], td["u"], td["v"], refr)
    ctx.ctuw_refra(gd, td["cg"], td["om"], td["wn"], refr, fail, float(cfg.idelpro), llcflcuroff=llcflcuroff)
    f3 = torch.full_like(td["f1"], -7.0)
    ctx.propags2_refra(td["f1"], f3, gd, td["cg"], td["om"], td["wn"], refr, float(cfg.idelpro), 0, n)
    torch.cuda.synchronize()
    out = dict(refr=refr.cpu().numpy(), fail=fail.cpu().numpy(), f3=f3.cpu().numpy(), wlat=gd["wlat"].cpu().numpy(),
               wcor=gd["wcor"].cpu().numpy())
    return ctx, gd, td, refr, out


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("irefra", [1, 2, 3])
def test_refraction_parity(api, prec, irefra):
    c = _case(prec, irefra)
    cfg, n, NANG, NR = c["cfg"], c["n"], c["cfg"].nang, c["cfg"].nfre_red
    dot, w, f3ref = _oracle_run(c, prec)
    assert w["NFAIL"] == 0 and np.all(w["CURMASK"] == 1)
    ctx, gd, td, refr, out = _device_run(api, c)
    eps = np.finfo(H.np_dtype(prec)).eps
    r = out["refr"].astype(float)
    # PROPDOT: THD(K) = THDD (IREFRA = 1) or THDC (2, 3); SDOT rebuilt from S0, OMDD as the kernel does
    thd_ref = (dot["THDD"] if irefra == 1 else dot["THDC"]).astype(float)
    sc = max(np.abs(thd_ref).max(), 1e-300)
    assert np.abs(r[:, :NANG] - thd_ref).max() < 16 * eps * sc
    if irefra >= 2:
        cg, om, wn = (c[k][:n, :NR].astype(float) for k in ("cg", "om", "wn"))
        sdot = (r[:, NANG:2 * NANG, None] * cg[:, None, :] + r[:, 2 * NANG + 2, None, None] * om[:, None, :]) * wn[:, None, :]
        ssc = np.abs(dot["SDOT"]).max()
        assert ssc > 0 and np.abs(sdot - dot["SDOT"].astype(float)).max() < 32 * eps * ssc
        assert np.array_equal(out["refr"][:, 2 * NANG], c["u"][:n]) and np.array_equal(out["refr"][:, 2 * NANG + 1], c["v"][:n])
    assert np.all(r[:, 2 * NANG + 3] == 1.0) and int(out["fail"].sum()) == 0
    assert np.array_equal(out["wlat"], w["WLAT"]) and np.array_equal(out["wcor"], w["WCOR"])
    # PROPAGS2
    f3 = out["f3"]
    fsc = np.abs(f3ref).max()
    assert np.abs(f3[:n, :, :NR].astype(float) - f3ref[:n, :, :NR].astype(float)).max() < 32 * eps * fsc
    assert np.array_equal(f3[:n, :, NR:], c["f1"][:n, :, NR:]) and np.all(f3[n] == -7.0)
    # the refraction terms matter in this case (the test would otherwise not exercise them)
    from oracle.oracle import Oracle
    o0 = Oracle(Config(nang=cfg.nang, nfre=cfg.nfre, nfre_red=cfg.nfre_red, idelpro=cfg.idelpro, idelt=cfg.idelt), prec)
    f3_0 = o0.propags2(c["g"], c["f1"], o0.ctu_weights(c["g"], c["cg"], float(cfg.idelpro)))
    assert np.abs(f3ref[:n] - f3_0[:n]).max() > 1e3 * eps * fsc
    # frequency sub-range (fast-wave sub-step): only M <= 5 of rows [3, n-2) rewritten
    f3b = torch.full_like(td["f1"], -7.0)
    ctx.propags2_refra(td["f1"], f3b, gd, td["cg"], td["om"], td["wn"], refr, float(cfg.idelpro), 3, n - 2, 1, 5, copy_rest=False)
    torch.cuda.synchronize()
    f3b = f3b.cpu().numpy()
    assert np.all(f3b[:3] == -7.0) and np.all(f3b[n - 2:] == -7.0) and np.all(f3b[3:n - 2, :, 5:] == -7.0)
    assert np.array_equal(f3b[3:n - 2, :, :5], f3[3:n - 2, :, :5])
    ctx.close()


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_current_cfl_failure_switches_refraction_off_locally(api, prec):
    """LLCFLCUROFF (ctuwdrv.F90:101-118): currents strong enough to break the CFL / weight-range checks at some points.
    The second CTUW call masks the current refraction and frequency shift of exactly those points; what still fails
    afterwards is reported (the reference aborts, ctuwdrv.F90:124-146)."""
    # depth + currents over rough bathymetry: the depth-gradient part of sigma-dot (OMDD, propdot.F90:126) pushes the
    # frequency-shift weights out of range at a few shallow points
    c = _case(prec, 3, delpro=900, cur_amp=6.0, smooth_depth=False)
    n, NANG = c["n"], c["cfg"].nang
    dot, w1, _ = _oracle_run(c, prec, llcflcuroff=False)
    assert w1["NFAIL"] > 0
    dot, w, f3ref = _oracle_run(c, prec, llcflcuroff=True)
    nmask = int((w["CURMASK"] == 0).sum())
    assert nmask == w1["NFAIL"] and 0 < nmask < n
    ctx, gd, td, refr, out = _device_run(api, c, llcflcuroff=True)
    assert np.array_equal(out["refr"][:, 2 * NANG + 3], w["CURMASK"])
    assert np.array_equal(out["fail"], w["FAIL"])
    ok = w["FAIL"] == 0
    eps = np.finfo(H.np_dtype(prec)).eps
    NR = c["cfg"].nfre_red
    d = np.abs(out["f3"][:n, :, :NR].astype(float) - f3ref[:n, :, :NR].astype(float))
    assert d[ok].max() < 32 * eps * np.abs(f3ref).max()
    ctx.close()
    # without the second call the flags are those of the first
    ctx, gd, td, refr, out = _device_run(api, c, llcflcuroff=False)
    assert np.array_equal(out["fail"], w1["FAIL"]) and np.all(out["refr"][:, 2 * NANG + 3] == 1)
    ctx.close()


def test_zero_currents_reduce_to_the_plain_scheme(api):
    """IREFRA = 2 with UCUR = VCUR = 0 everywhere: no upwind switch, no current refraction, no frequency shift -- the
    general branch then differs from the IREFRA = 0 stencil by its summation order only."""
    c = _case("dp", 2, cur_amp=0.0)
    n, NR = c["n"], c["cfg"].nfre_red
    ctx, gd, td, refr, out = _device_run(api, c)
    f3plain = torch.full_like(td["f1"], -7.0)
    ctx.propags2_otf(td["f1"], f3plain, gd, td["cg"], float(c["cfg"].idelpro), 0, n)
    torch.cuda.synchronize()
    a, b = out["f3"][:n, :, :NR], f3plain.cpu().numpy()[:n, :, :NR]
    assert np.abs(a - b).max() < 8 * np.finfo(np.float64).eps * np.abs(b).max()
    ctx.close()


class _LateHalo:
    """Stands in for HaloExchange on emulated ranks: the halo rows hold NaN until `finish`, so an interior pass that read
    one would poison its result."""

    def __init__(self, m, source):
        self.m, self.source = m, source

    def start(self, fl):
        fl[self.m.n: self.m.n + self.m.dom.nh] = float("nan")
        return []

    def finish(self, reqs):
        glob = self.source()
        hg = torch.from_numpy(np.asarray(self.m.dom.halo_global, dtype=np.int64)).to(glob.device)
        self.m.fl1[self.m.n: self.m.n + self.m.dom.nh] = glob[hg]

    def __call__(self, fl):
        self.finish(self.start(fl))


@pytest.mark.parametrize("irefra,weights", [(0, "otf"), (0, "stored"), (2, "otf"), (3, "otf")])
def test_decomposed_step_with_overlapped_exchange_is_bit_identical(api, irefra, weights):
    """Wamintgr.propag on 3 emulated ranks -- halo exchange posted, interior rows advected, exchange completed, the two
    ends of the band advected (what `bench.py --gpus N` runs; with currents the halo rows of DEPTH/UCUR/VCUR come from the
    global fields as PROENVHALO exchanges them) -- reproduces the single-domain run bit for bit."""
    from ecwam_amd import grid as G
    from ecwam_amd.wamintgr import Wamintgr

    cfg = Config(nang=12, nfre=36, nfre_red=28, idelt=600, idelpro=600, irefra=irefra)
    g = G.build_grid(20, mask="continents")
    ref = Wamintgr(cfg, g, "sp", weights=weights)
    ref.init_synthetic(seed=11)
    assert ref.build_weights() == 0
    nr = 3
    parts = []
    snapshot = {}
    for r in range(nr):
        m = Wamintgr(cfg, g, "sp", rank=r, nranks=nr, weights=weights)
        m.init_synthetic(seed=11)
        assert m.build_weights() == 0
        a, b = m.interior
        assert 0 < a < b < m.n or r in (0, nr - 1)                 # a band in the middle has halo readers at both ends
        assert b - a > 0.5 * m.n
        m.halo = _LateHalo(m, lambda: snapshot["glob"])
        parts.append(m)
    if irefra:
        assert float(ref.refr[:, 2 * cfg.nang].abs().max()) > 0.1     # currents present
    for _ in range(2):
        ref.step()
        snapshot["glob"] = torch.cat([m.fl1[: m.n] for m in parts])   # owned rows of every rank at the old time level
        for m in parts:
            m.step()
    torch.cuda.synchronize()
    got = torch.cat([m.fl1[: m.n] for m in parts]).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got, ref.fl1[: g.nsea].cpu().numpy())
    for m in parts + [ref]:
        m.ctx.close()


@pytest.mark.parametrize("prec", ["sp", "dp"])
def test_proenvhalo_on_the_device_equals_the_host_assembly(api, prec):
    """PROENVHALO (proenvhalo.F90:63-107) on the device: owned rows packed from the device-resident fields, halo rows exchanged as rows of
    3 NFRE + 3 reals (here: copied from the owners' packed rows, what MPEXCHNG delivers), land slot from WVPRPT_LAND -- the extended DEPTH /
    UCUR / VCUR / OMOSNH2KD / WAVNUM / CGROUP arrays of every rank equal those assembled on the host from the global fields bit for bit, and
    so do the dot terms and CFL flags built from them."""
    from ecwam_amd import grid as G
    from ecwam_amd.wamintgr import Wamintgr

    cfg = Config(nang=12, nfre=36, nfre_red=28, idelt=600, idelpro=600, irefra=3)
    g = G.build_grid(20, mask="continents")
    nr = 3
    host = [Wamintgr(cfg, g, prec, rank=r, nranks=nr) for r in range(nr)]
    dev = [Wamintgr(cfg, g, prec, rank=r, nranks=nr) for r in range(nr)]
    for m in host + dev:
        m.init_synthetic(seed=11)                       # extended rows assembled on the host from the global synthetic fields
    names = ("depth_ext", "u_ext", "v_ext", "omosnh2kd_ext", "wavnum_ext", "cgroup_ext")
    packed = []
    for m, h in zip(dev, host):
        n = m.n
        own = [getattr(h, k)[:n].clone() for k in names[:4]]
        for k in names:                                  # poison what PROENVHALO has to rebuild
            getattr(m, k).fill_(float("nan"))
        packed.append(m.proenvhalo_pack(*own))
    owned = torch.cat([b[: m.n] for b, m in zip(packed, dev)])            # rows of all ranks in global point order
    for m, b in zip(dev, packed):
        hg = torch.from_numpy(np.asarray(m.dom.halo_global, dtype=np.int64)).to(owned.device)
        b[m.n: m.n + m.dom.nh] = owned[hg]                                 # MPEXCHNG(BUFFER_EXT, 3*NFRE_RED+5, 1, 1)
        m.proenvhalo_unpack()
    torch.cuda.synchronize()
    for m, h in zip(dev, host):
        for k in names:
            assert torch.equal(getattr(m, k), getattr(h, k)), k
        assert m.weights_ready is False
        assert m.build_weights() == h.build_weights()
        assert torch.equal(m.refr, h.refr) and torch.equal(m.cflfail, h.cflfail)
    assert float(host[1].u_ext.abs().max()) > 0.1 and host[1].dom.nh > 0
    for m in host + dev:
        m.ctx.close()


@pytest.mark.parametrize("prec", ["dp", "sp"])
@pytest.mark.parametrize("irefra", [0, 2])
def test_subgrid_obstructions(api, prec, irefra):
    """LSUBGRID (ctuw.F90:703-733): the transmission coefficients OBSLAT/OBSLON/OBSCOR scale the space weights of the
    neighbours after the checks; SUMWN is left alone (blocked energy is lost).  Stored weights, on-the-fly weights and the
    refraction kernel against the oracle; switching the table off restores the plain result."""
    from ecwam_amd import synthetic as syn

    c = _case(prec, irefra)
    cfg, n, NR, g = c["cfg"], c["n"], c["cfg"].nfre_red, c["g"]
    obs = syn.obstructions(g, cfg.nfre).astype(H.np_dtype(prec))
    from oracle.oracle import Oracle
    o = Oracle(cfg, prec)
    o.set_obstructions(obs)
    if irefra:
        dot = o.propdot(g, irefra, c["dep"], c["u"], c["v"], c["wn"], c["cg"], c["om"])
        w = o.ctu_weights_gen(g, irefra, c["cg"], c["om"], c["u"], c["v"], dot, float(cfg.idelpro))
        f3ref = o.propags2_gen(g, c["f1"], w)
    else:
        w = o.ctu_weights(g, c["cg"], float(cfg.idelpro))
        f3ref = o.propags2(g, c["f1"], w)
    assert w["NFAIL"] == 0
    o.set_obstructions(None)
    w0 = o.ctu_weights(g, c["cg"], float(cfg.idelpro))
    assert np.array_equal(w0["SUMWN"], w["SUMWN"]) if not irefra else True
    assert (w["WLONN"] <= o.ctu_weights_gen(g, irefra, c["cg"], c["om"], c["u"], c["v"], dot, float(cfg.idelpro))["WLONN"]).all() if irefra else (w["WLONN"] <= w0["WLONN"]).all()
    ctx = api.HipContext(c["t"])
    dev = ctx.device
    gd = api.grid_to_device(g, ctx.dtype, dev)
    td = {k: torch.from_numpy(c[k]).to(dev) for k in ("cg", "om", "wn", "dep", "u", "v", "f1")}
    tobs = torch.from_numpy(obs).to(dev)
    ctx.set_obstructions(tobs)
    fail = torch.zeros(n, dtype=torch.int32, device=dev)
    eps = np.finfo(H.np_dtype(prec)).eps
    fsc = np.abs(f3ref).max()
    if irefra:
        refr = torch.zeros((n, 2 * cfg.nang + 5), dtype=ctx.dtype, device=dev)
        ctx.propdot(gd, td["dep"], td["u"], td["v"], refr)
        ctx.ctuw_refra(gd, td["cg"], td["om"], td["wn"], refr, fail, float(cfg.idelpro))
        f3 = torch.full_like(td["f1"], -7.0)
        ctx.propags2_refra(td["f1"], f3, gd, td["cg"], td["om"], td["wn"], refr, float(cfg.idelpro), 0, n)
        torch.cuda.synchronize()
        got = f3.cpu().numpy()
        assert np.abs(got[:n, :, :NR].astype(float) - f3ref[:n, :, :NR].astype(float)).max() < 32 * eps * fsc
        ctx.set_obstructions(None)
        f3b = torch.full_like(td["f1"], -7.0)
        ctx.propags2_refra(td["f1"], f3b, gd, td["cg"], td["om"], td["wn"], refr, float(cfg.idelpro), 0, n)
        torch.cuda.synchronize()
        assert np.abs(f3b.cpu().numpy()[:n] - got[:n]).max() > 1e3 * eps * fsc       # the table acted
    else:
        wdev = torch.zeros((n, 8, cfg.nang * NR), dtype=ctx.dtype, device=dev)
        ctx.ctuw(gd, td["cg"], wdev, fail, float(cfg.idelpro))
        f3s = torch.full_like(td["f1"], -7.0)
        ctx.propags2(td["f1"], f3s, gd["klon"], gd["klat"], gd["kcor"], wdev, 0, n)
        f3o = torch.full_like(td["f1"], -7.0)
        ctx.propags2_otf(td["f1"], f3o, gd, td["cg"], float(cfg.idelpro), 0, n)
        f3p = torch.full_like(td["f1"], -7.0)
        ctx.propags2_otf(td["f1"], f3p, gd, td["cg"], float(cfg.idelpro), 3, n - 2, 1, 5, copy_rest=False)   # scalar kernel variant
        torch.cuda.synchronize()
        got = f3o.cpu().numpy()
        H.assert_same_advection(got, f3s.cpu().numpy(), eps, fsc)                    # stored and on-the-fly weights
        assert np.array_equal(f3p.cpu().numpy()[3:n - 2, :, :5], got[3:n - 2, :, :5])
        assert np.abs(got[:n, :, :NR].astype(float) - f3ref[:n, :, :NR].astype(float)).max() < 16 * eps * fsc
        ctx.set_obstructions(None)
        f3b = torch.full_like(td["f1"], -7.0)
        ctx.propags2_otf(td["f1"], f3b, gd, td["cg"], float(cfg.idelpro), 0, n)
        torch.cuda.synchronize()
        plain = o.propags2(g, c["f1"], w0)
        assert np.abs(f3b.cpu().numpy()[:n, :, :NR].astype(float) - plain[:n, :, :NR].astype(float)).max() < 16 * eps * fsc
        assert np.abs(f3b.cpu().numpy()[:n] - got[:n]).max() > 1e3 * eps * fsc
        # energy only leaves: the obstructed field never exceeds the open one
        assert (got[:n, :, :NR] <= f3b.cpu().numpy()[:n, :, :NR] * (1 + 8 * eps) + 1e-300).all()
    assert int(fail.sum()) == 0
    ctx.close()


def test_decomposed_step_with_obstructions_is_bit_identical(api):
    from ecwam_amd import grid as G, synthetic as syn
    from ecwam_amd.wamintgr import Wamintgr

    cfg = Config(nang=12, nfre=36, nfre_red=28, idelt=600, idelpro=600)
    g = G.build_grid(20, mask="continents")
    obs = syn.obstructions(g, cfg.nfre)
    ref = Wamintgr(cfg, g, "sp")
    ref.init_synthetic(seed=11)
    ref.set_obstructions(obs)
    assert ref.build_weights() == 0
    plain = Wamintgr(cfg, g, "sp")
    plain.init_synthetic(seed=11)
    parts, snapshot = [], {}
    for r in range(3):
        m = Wamintgr(cfg, g, "sp", rank=r, nranks=3)
        m.init_synthetic(seed=11)
        m.set_obstructions(obs)
        assert m.build_weights() == 0
        m.halo = _LateHalo(m, lambda: snapshot["glob"])
        parts.append(m)
    for _ in range(2):
        ref.step()
        plain.step()
        snapshot["glob"] = torch.cat([m.fl1[: m.n] for m in parts])
        for m in parts:
            m.step()
    torch.cuda.synchronize()
    got = torch.cat([m.fl1[: m.n] for m in parts]).cpu().numpy()
    assert np.array_equal(got, ref.fl1[: g.nsea].cpu().numpy())
    assert not np.array_equal(got, plain.fl1[: g.nsea].cpu().numpy())
    for m in parts + [ref, plain]:
        m.ctx.close()
