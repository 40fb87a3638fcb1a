"""Known-answer checks of the hot path against the PUBLISHED equations (IFS documentation Part VII; Janssen 1991; Ardhuin et al. 2010;
Kenyon 1969; the corner-transport upstream scheme), written once and run on both implementations: `run` is either the CPU oracle or the
HIP library through the C ABI (tests/test_known_answers.py, tests/test_gpu_known_answers.py).  Nothing here calls one implementation to
check the other: every expectation is computed from the inputs (and, where a formula relates two outputs, from the run's own outputs)
with numpy.  Test infrastructure only.

The reference holds no golden vector for this path: these checks tie BOTH restatements to the physics they claim."""
from __future__ import annotations

import numpy as np

import harness as H
from ecwam_amd import synthetic as syn
from ecwam_amd.tables import Config, Tables

G_ = 9.806


def flat_case(prec: str, level: float, n: int = 48, u10: float = 12.0, idelt: int = 450, seed: int = 3, f0: float = 0.2):
    """Deep water, no ice, no convective gustiness, a spectrum without directional structure: `level` up to f0, falling as f^-5 above it
    (the stress the short waves would support under a flat spectrum shelters them, sinput_ard.F90:356-372: a real tail keeps that
    out of the picture).  Where it lies far above the noise floor FLMIN = 1e-5 it is also far below the saturation threshold of the
    whitecapping term and far too small for the cubic DIA to matter: the only source term left is the wind input, linear in the spectrum."""
    cfg = Config(nang=24, nfre=36, nfre_red=36, idelt=idelt, idelpro=idelt)
    case = H.make_point_case(n, cfg, prec, seed=seed)
    dt = H.np_dtype(prec)
    t = case["tables"]
    depth = np.full(n, 998.999)
    case["params"]["DEPTH"] = depth
    case["props"] = syn.depth_props(depth, t, dt)
    case["ENV"] = np.stack([case["props"]["EMAXDPT"], depth.astype(dt)], 1)
    ff = case["FF"]
    rng = np.random.default_rng(seed)
    ff[:, 0] = 1.225                                        # AIRD
    ff[:, 1] = rng.uniform(0, 2 * np.pi, n)                 # WDWAVE
    ff[:, 2] = 0.0                                          # CICOVER
    ff[:, 3] = u10 * rng.uniform(0.6, 1.4, n)               # WSWAVE
    ff[:, 4] = 0.0                                          # WSTAR: SIG_N = 0 (wsigstar.F90:68: no background gustiness), both gust states = u*
    ff[:, 5:7] = 0.0
    ff[:, 7] = np.sqrt(1.2e-3) * ff[:, 3]                   # UFRIC first guess
    ff[:, 8] = 0.0                                          # TAUW
    ff[:, 9] = ff[:, 1]
    ff[:, 10] = 1e-4
    ff[:, 11] = 1e-4
    ff[:, 12] = 0.0185
    ff[:, 13] = 0.0
    fr = np.asarray(t.FR, float)
    shape = level * np.minimum(1.0, (f0 / fr) ** 5)
    case["FL1"] = np.broadcast_to(shape[None, None, :], (n, cfg.nang, cfg.nfre)).astype(dt).copy()
    return case


def janssen_growth_rate(t, ustar, z0, aird, wdwave, wavnum, cinv):
    """gamma(k, theta) [1/s] = sigma eps (betamax / kappa^2) mu ln^4(mu) x^2, mu = k z0 exp(kappa / ((u*/c + z_alpha) cos)) <= 1,
    x = (u*/c) cos(theta - phi), eps = rho_a / rho_w (Janssen 1991; IFS documentation Part VII eq. 3.9-3.11; sinput_ard.F90:422-433).
    Shapes: per point [n], wavnum / cinv [n][NFRE]; result [n][NANG][NFRE]."""
    th = np.asarray(t.TH, float)
    sig = np.asarray(t.ZPIFR, float)[None, None, :]
    cosd = np.cos(th[None, :, None] - wdwave[:, None, None])
    uoc = (ustar[:, None] * cinv)[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        lnmu = np.log(wavnum * z0[:, None])[:, None, :] + float(t.XKAPPA) / (cosd * (uoc + float(t.ZALP)))
        g = np.exp(lnmu) * lnmu ** 4 * (uoc * cosd) ** 2
    g = np.where((cosd > 0.01) & (lnmu < 0), g, 0.0)
    eps = (np.maximum(aird, 1.0) / float(t.ROWATER))[:, None, None]
    return sig * eps * float(t.BETAMAXOXKAPPA2) * g


def check_wind_input_growth_rate(run, prec):
    """SINPUT_ARD through the whole of IMPLSCH: on the flat low spectrum every bin changes by DELT gamma F (explicit where the source is
    positive, implsch.F90:384-386), with gamma the published growth rate evaluated from the run's OWN friction velocity and roughness.
    Twice the spectrum changes twice as much (linearity: the DIA and the dissipation are out of the picture).  The swell-damping terms of
    Ardhuin et al. 2010 (viscous + turbulent, sinput_ard.F90:179-271) act on every bin: O(1e-5 / s), bounded here instead of restated."""
    lev = 2.0e-2
    case = flat_case(prec, lev)
    t, cfg = case["tables"], case["cfg"]
    out = run.implsch(case)
    case2 = flat_case(prec, 2 * lev)
    out2 = run.implsch(case2)
    n = case["n"]
    pr = case["props"]
    us, z0 = out["FF"][:, 7].astype(float), out["FF"][:, 10].astype(float)
    gam = janssen_growth_rate(t, us, z0, case["FF"][:, 0].astype(float), case["FF"][:, 1].astype(float), pr["WAVNUM"].astype(float),
                              pr["CINV"].astype(float))
    delt = float(cfg.idelt)
    f_in, f_in2 = case["FL1"].astype(float), case2["FL1"].astype(float)
    rate = (out["FL1"].astype(float) - f_in) / (delt * f_in)
    rate2 = (out2["FL1"].astype(float) - f_in2) / (delt * f_in2)
    # bins the limiter leaves alone (|dF| <= u* fm 5e-7 g / f^4 DELT, Hersbach & Janssen 1999: implsch.F90:386), below the cut-off MIJ
    fr = np.asarray(t.FR, float)
    m_idx = np.arange(cfg.nfre)[None, None, :]
    below = np.broadcast_to(m_idx < (out["MIJ"][:, None, None] - 1), case["FL1"].shape)
    cofrm4 = np.asarray(t.COFRM4, float)[None, None, :]       # the limiter's 5e-7 g / f^4 (table of the model set-up)
    sel = below & (gam * delt > 0.02) & (gam * delt < 0.5) & (gam * delt * f_in < 0.25 * us[:, None, None] * 0.1 * cofrm4 * delt) & (f_in > 1e-4)
    assert sel.sum() > 20 * n, int(sel.sum())
    # the swell damping of Ardhuin et al. 2010 (viscous + turbulent, sinput_ard.F90:179-271,454-461) acts on every bin and is linear in
    # cos(theta - phi) at a frequency: it is read off the run's own opposing-wind bins (no growth there) and continued to the others
    th_ = np.asarray(t.TH, float)
    cosd = np.cos(th_[None, :, None] - case["FF"][:, 1].astype(float)[:, None, None]) * np.ones_like(rate)
    wgt = (cosd < -0.05).astype(float)
    sw, sc, scc = wgt.sum(1), (wgt * cosd).sum(1), (wgt * cosd * cosd).sum(1)
    sy, scy = (wgt * rate).sum(1), (wgt * cosd * rate).sum(1)
    det = sw * scc - sc * sc
    a_fit, b_fit = (scc * sy - sc * scy) / det, (sw * scy - sc * sy) / det
    damp = a_fit[:, None, :] + b_fit[:, None, :] * cosd
    assert np.all(damp[below] <= 0.0) and np.all(damp[below] > -2e-4)
    err = np.abs(rate - damp - gam)[sel]
    tol = 0.02 * gam[sel] + 1e-7      # 2 %: the sheltering of the short waves by the long ones' stress (TAUWSHELTER) lowers u* by a per cent
    assert np.all(err < tol), (float((err / tol).max()), float(gam[sel].min()), float(gam[sel].max()))
    # twice the spectrum: the same relative change up to what the doubled wave stress does to u* and z0 (a per cent): nothing cubic
    # (DIA) or thresholded (whitecapping) is at work
    lin = np.abs(rate2 - rate)[sel] / gam[sel]
    assert np.max(lin) < 0.15 and np.median(lin) < 0.01, (float(np.max(lin)), float(np.median(lin)))      # (the onset of growth is steep in u*)
    # growth only within 90 degrees of the wind; opposing bins are damped, and only slightly
    th = np.asarray(t.TH, float)
    opp = np.cos(th[None, :, None] - case["FF"][:, 1].astype(float)[:, None, None]) < -0.2
    assert np.all(rate[opp & below] <= 0.0) and np.all(rate[opp & below] > -2e-4)
    return dict(case=case, out=out, gam=gam, damp=damp)


def check_momentum_balance(run, prec, grown):
    """WNFLUXES: the stress the ocean receives is the atmospheric stress minus what the waves keep, tau_oc = tau_a - rho_w g INT S / c
    (k / k) df dtheta (wnfluxes.F90:147-190,267-271; IFS documentation Part VII eq. 3.31), up to the cut-off frequency MIJ with half a
    bin there (frcutindex.F90:98-107).  With the pure wind input of the flat spectrum S = gamma F is known analytically."""
    case, out, gam = grown["case"], grown["out"], grown["gam"] + grown["damp"]      # S / F = growth + swell damping
    t, cfg = case["tables"], case["cfg"]
    th = np.asarray(t.TH, float)
    f_in = case["FL1"].astype(float)
    us = out["FF"][:, 7].astype(float)
    aird = case["FF"][:, 0].astype(float)
    wd = case["FF"][:, 1].astype(float)
    tau = aird * np.maximum(us ** 2, float(t.EPSUS))
    intf = out["INTF"].astype(float)
    tol = 2e-5 if prec == "sp" else 1e-12
    assert np.allclose(intf[:, 5], tau * np.sin(wd), rtol=tol, atol=tol * tau.max()) and np.allclose(intf[:, 6], tau * np.cos(wd), rtol=tol, atol=tol * tau.max())
    mij = out["MIJ"]
    w = np.clip(mij[:, None] + np.where(mij == cfg.nfre, 1.0, 0.5)[:, None] - (np.arange(cfg.nfre)[None, :] + 1), 0.0, 1.0)
    dfim = np.asarray(t.DFIM, float)[None, :]
    kern = float(t.ROWATER) * G_ * dfim * w * case["props"]["CINV"].astype(float)          # [n][NFRE]
    sx = (kern[:, None, :] * gam * f_in * np.sin(th)[None, :, None]).sum((1, 2))
    sy = (kern[:, None, :] * gam * f_in * np.cos(th)[None, :, None]).sum((1, 2))
    wave = np.hypot(sx, sy)
    assert np.all(wave > 1e-4 * tau)       # the check has something to see
    ex, ey = intf[:, 7] - (intf[:, 5] - sx), intf[:, 8] - (intf[:, 6] - sy)
    # 3 %: the analytic growth rate leaves out the swell damping and the sheltering of the short waves (both per cent effects)
    assert np.all(np.hypot(ex, ey) < 0.03 * wave + tol * tau), float((np.hypot(ex, ey) / wave).max())
    tauoc = np.clip(np.hypot(intf[:, 7], intf[:, 8]) / tau, float(t.TAUOCMIN), float(t.TAUOCMAX))
    assert np.allclose(intf[:, 9], tauoc, rtol=10 * tol)
    xn = aird * np.maximum(us ** 3, float(t.EPSUS) ** 1.5)
    assert np.allclose(intf[:, 12], intf[:, 13] * xn, rtol=10 * tol) and np.all(intf[:, 13] <= float(t.PHIEPSMAX)) and np.all(intf[:, 13] >= float(t.PHIEPSMIN))


def check_tail_and_stokes_drift(run, prec):
    """IMPHFTAIL: above the cut-off frequency the spectrum continues as f^-5 (deep water: (k^3 c_g)(MIJ) / (k^3 c_g)(M) = (f_MIJ / f_M)^5,
    imphftail.F90:77-88) down to the noise floor FLMIN max(0, cos(theta - phi))^2.  STOKESDRIFT: the surface Stokes drift of the new
    spectrum is (16 pi^3 / g) INT f^3 F (sin, cos) df dtheta in deep water (Kenyon 1969; stokesdrift.F90:90-130) -- here by the midpoint
    rule on the logarithmic frequency grid with the f^-5 tail added analytically, against the reference's Simpson sum."""
    cfg = Config(nang=24, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    n = 96
    case = H.make_point_case(n, cfg, prec, seed=17)
    dt = H.np_dtype(prec)
    t = case["tables"]
    depth = np.full(n, 998.999)
    case["props"] = syn.depth_props(depth, t, dt)
    case["ENV"] = np.stack([case["props"]["EMAXDPT"], depth.astype(dt)], 1)
    case["FF"][:, 2] = 0.0
    case["FF"][:, 13] = 0.0
    out = run.implsch(case)
    fl = out["FL1"].astype(float)
    fr = np.asarray(t.FR, float)
    th = np.asarray(t.TH, float)
    mij = out["MIJ"]
    wd = case["FF"][:, 1].astype(float)
    cosd = np.cos(th[None, :] - wd[:, None])
    flm = float(t.FLMIN) * np.maximum(0.0, cosd) ** 2
    # a bin on the floor carries the rounding of TH - WDWAVE in the working precision: d(COS**2) <= 2 |COS| d(angle), d(angle) a few ulp of 2 pi
    # (seen with other seeds, ECWAM_TEST_SEED_OFFSET: 5.8e-5 of the floor where COS = 0.004 in single precision)
    flm_atol = float(t.FLMIN) * 2.0 * np.abs(cosd) * (2e-6 if prec == "sp" else 4e-15)
    seen = 0
    for i in range(n):
        mi = int(mij[i])
        if mi >= cfg.nfre:
            continue
        kd = case["props"]["WAVNUM"][i].astype(float) * depth[i]
        for m in range(mi, cfg.nfre):
            want = np.maximum(fl[i, :, mi - 1] * (fr[mi - 1] / fr[m]) ** 5, flm[i])
            rtol = (1e-5 if prec == "sp" else 1e-12) if kd[mi - 1] > 10 else 5e-4          # AKI's tolerance below k d = 10 (aki.F90:71-91)
            assert np.all(np.abs(fl[i, :, m] - want) <= rtol * np.abs(want) + flm_atol[i]), (i, m)
            seen += 1
    assert seen > 100
    # Stokes drift of the run's own new spectrum
    dth = 2 * np.pi / cfg.nang
    h = np.log(float(t.FRATIO))
    e1 = (fl * np.sin(th)[None, :, None]).sum(1) * dth
    e2 = (fl * np.cos(th)[None, :, None]).sum(1) * dth
    c = 16 * np.pi ** 3 / G_
    us = c * ((fr[None, :] ** 4 * e1).sum(1) * h + fr[-1] ** 4 * e1[:, -1] * (1.0 - 0.5 * h))      # midpoint cells end at f_N sqrt(ratio); tail beyond
    vs = c * ((fr[None, :] ** 4 * e2).sum(1) * h + fr[-1] ** 4 * e2[:, -1] * (1.0 - 0.5 * h))
    got_u, got_v = out["INTF"][:, 2].astype(float), out["INTF"][:, 3].astype(float)
    mag = np.hypot(us, vs)
    assert np.all(mag < 1.4) and mag.max() > 0.02
    dev = np.hypot(got_u - us, got_v - vs) / mag      # two quadratures of a spectrum with a sharp peak on 36 geometric frequencies
    assert np.all(dev < 0.05) and np.median(dev) < 0.02, (float(dev.max()), float(np.median(dev)))


def check_ctu_single_bin(run, prec):
    """PROPAGS2 / CTUW, one time step of a single bin on a uniform-depth aqua grid: the point keeps (1 - cx)(1 - cy) - c_theta of it, its
    upstream neighbour in the row receives cx (1 - cy), the two neighbouring directions c_theta(+-) -- with the Courant numbers of the
    corner-transport upstream scheme on the sphere, cx = c_g |sin theta| dt / (R cos(phi) dlambda), cy = c_g |cos theta| dt / (R dphi),
    and the great-circle turning rate d theta / dt = (c_g / R) tan(phi) sin(theta) (theta clockwise from north) (ctuw.F90:160-275,413-484; propags2.F90:107-116).
    The scheme moves energy, it does not create it: the field sums to one again (to the order of the grid's non-uniformity)."""
    from ecwam_amd import grid as G

    g = G.build_grid(48, mask="aqua")
    cfg = Config(nang=36, nfre=36, nfre_red=36, idelpro=900)
    dt = H.np_dtype(prec)
    t = Tables(cfg, dt)
    props = syn.depth_props(np.array([998.999]), t, dt)
    cg_ext = np.repeat(props["CGROUP"], g.nsea + 1, axis=0).astype(dt)
    R = float(t.R)
    th = np.asarray(t.TH, float)
    dth = 2 * np.pi / cfg.nang
    delpro = float(cfg.idelpro)
    checked = 0
    for row, K, M in ((60, 7, 3), (30, 20, 10), (70, 33, 0), (48, 14, 6)):
        a, b = int(g.row_start[row]), int(g.row_start[row + 1])
        ij0 = (a + b) // 2
        f1 = np.zeros((g.nsea + 1, cfg.nang, cfg.nfre), dt)
        f1[ij0, K, M] = 1.0
        f3 = run.propags2(g, cfg, t, f1, cg_ext, delpro).astype(float)
        cg = float(props["CGROUP"][0, M])
        phi = np.arcsin(float(g.sinph[row]))
        dlam = np.deg2rad(float(g.zdello[row]))
        dphi = np.deg2rad(float(g.xdella))
        cx = cg * abs(np.sin(th[K])) * delpro / (R * np.cos(phi) * dlam)
        cy = cg * abs(np.cos(th[K])) * delpro / (R * dphi)
        kp, km = (K + 1) % cfg.nang, (K - 1) % cfg.nang
        rate = lambda k2: (cg / R) * np.tan(phi) * 0.5 * (np.sin(th[K]) + np.sin(th[k2])) * delpro / dth
        cp, cm = max(rate(kp), 0.0), max(-rate(km), 0.0)
        lost = 1.0 - f3[ij0, K, M]
        assert 0.01 < lost < 0.9
        want_lost = cx + cy - cx * cy + cp + cm
        assert abs(lost - want_lost) < 0.03 * want_lost, (row, K, lost, want_lost)
        down = ij0 + 1 if np.sin(th[K]) > 0 else ij0 - 1       # the point the energy moves to along the row (sea points are numbered west -> east)
        assert abs(f3[down, K, M] - cx * (1 - cy)) < 0.04 * cx, (row, K, f3[down, K, M], cx * (1 - cy))
        assert abs(f3[ij0, kp, M] - cp) < 0.03 * max(cp, cm) + 1e-7 and abs(f3[ij0, km, M] - cm) < 0.03 * max(cp, cm) + 1e-7
        tot = f3[: g.nsea, :, M].sum()
        assert f3.min() >= 0.0 and abs(tot - 1.0) < 0.05 * lost, (row, K, tot)
        assert not f3[: g.nsea, :, np.arange(cfg.nfre) != M].any()
        checked += 1
    assert checked == 4


class OracleRun:
    """The CPU oracle behind the interface the checks use."""

    def __init__(self, prec):
        self.prec = prec
        self._o = {}

    def _oracle(self, cfg):
        from oracle.oracle import Oracle
        key = tuple(sorted((k, v) for k, v in vars(cfg).items() if isinstance(v, (int, float, bool))))
        if key not in self._o:
            self._o[key] = Oracle(cfg, self.prec)
        return self._o[key]

    def implsch(self, case):
        return H.oracle_implsch(case, self._oracle(case["cfg"]))

    def propags2(self, g, cfg, t, f1, cg_ext, delpro):
        o = self._oracle(cfg)
        return o.propags2(g, f1, o.ctu_weights(g, cg_ext, delpro))


class DeviceRun:
    """The HIP library through the C ABI (ecwam_amd.api.HipContext)."""

    def __init__(self, prec):
        self.prec = prec

    def implsch(self, case):
        from ecwam_amd import api
        ctx = api.HipContext(case["tables"])
        try:
            return H.gpu_implsch(case, ctx)
        finally:
            ctx.close()

    def propags2(self, g, cfg, t, f1, cg_ext, delpro):
        import torch
        from ecwam_amd import api
        ctx = api.HipContext(t)
        try:
            dev = ctx.device
            gd = api.grid_to_device(g, ctx.dtype, dev)
            cg = torch.from_numpy(cg_ext).to(dev)
            fail = torch.zeros(g.nsea, dtype=torch.int32, device=dev)
            ctx.ctuw(gd, cg, None, fail, delpro)          # CTUWINI's snapping of WLAT / WCOR + the CFL checks
            assert int(fail.sum()) == 0
            a = torch.from_numpy(f1).to(dev)
            b = torch.zeros_like(a)
            ctx.propags2_otf(a, b, gd, cg, delpro, 0, g.nsea)
            torch.cuda.synchronize()
            return b.cpu().numpy()
        finally:
            ctx.close()
