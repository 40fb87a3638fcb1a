"""Known-answer tests of the oracle against the PUBLISHED equations the reference implements (IFS documentation Part VII
"ECMWF wave model"; Janssen 1991; Hasselmann et al. 1985; Ardhuin et al. 2010), independent of the reference's source text.
The reference holds no golden vector for this path: these checks are what ties the restatement to the
physics it claims -- they do not replace a comparison with the reference's own output.
"""
import ctypes as C

import numpy as np
import pytest

import harness as H
from ecwam_amd import synthetic as syn
from ecwam_amd.tables import Config, Tables


def _oracle(cfg, prec="dp"):
    from oracle.oracle import Oracle

    return Oracle(cfg, prec)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_dispersion_relation_and_group_velocity():
    """DEPTHPRPT / AKI: omega^2 = g k tanh(k d); cg = d omega / d k = 0.5 c (1 + 2kd / sinh 2kd); deep-water limits."""
    cfg = Config(nang=12, nfre=36, nfre_red=25)
    t = Tables(cfg, np.float64)
    o = _oracle(cfg)
    d = np.array([2.0, 5.0, 17.0, 60.0, 250.0, 998.0])
    pr = o.depthprpt(d)
    om = 2 * np.pi * np.asarray(t.FR, float)[None, :]
    k = pr["WAVNUM"]
    g = 9.806
    kd = k * d[:, None]
    shallow = kd <= 10.0                       # the reference switches to the deep-water forms beyond k d = 10 (depthprpt.F90)
    assert np.max(np.abs(om ** 2 - g * k * np.tanh(kd))[shallow] / (om ** 2 * np.ones_like(k))[shallow]) < 2e-4   # AKI tolerance EBS = 1e-4
    assert np.max(np.abs(k[~shallow] - (om ** 2 / g * np.ones_like(k))[~shallow]) / k[~shallow]) < 1e-6
    cg_exact = 0.5 * (om / k) * (1 + 2 * kd / np.sinh(np.minimum(2 * kd, 700)))
    assert np.max(np.abs(pr["CGROUP"] - cg_exact)[shallow] / cg_exact[shallow]) < 2e-4
    assert np.allclose(pr["CGROUP"][~shallow], (g / (2 * om) * np.ones_like(k))[~shallow], rtol=1e-12)
    assert np.allclose(pr["CINV"], k / om, rtol=1e-13)


def test_dia_conserves_energy_and_has_the_three_lobe_shape():
    """SNONLIN (discrete interaction approximation, Hasselmann et al. 1985): the resonant quadruplets redistribute energy --
    the frequency-direction integral of S_nl is small against the integral of |S_nl| (it vanishes in the continuum; the
    bilinear interpolation onto the grid leaves O(1e-2)), and for a JONSWAP spectrum the one-dimensional transfer is
    positive on the forward face and at the peak, negative above it, positive again in the tail ("plus-minus-plus")."""
    cfg = Config(nang=24, nfre=36, nfre_red=36)
    t = Tables(cfg, np.float64)
    o = _oracle(cfg)
    fp = 0.12
    fl = syn.jonswap_spectra(t.FR, t.TH, np.array([fp]), np.array([1.0]), np.float64)[0].copy()
    pr = o.depthprpt(np.array([998.0]))
    sl, fld = np.zeros_like(fl), np.zeros_like(fl)
    o.lib.ora_snonlin(_p(fl), C.c_double(998.0), C.c_double(float(pr["WAVNUM"][0, 8])), _p(pr["WAVNUM"][0].copy()), _p(sl), _p(fld))
    dfim = np.asarray(t.DFIM, float)
    s1d = sl.sum(0)                                                            # S(f) summed over direction
    tot, mag = (sl * dfim[None, :]).sum(), (np.abs(sl) * dfim[None, :]).sum()
    assert mag > 0 and abs(tot) < 0.03 * mag
    fr = np.asarray(t.FR, float)
    ipk = int(np.argmin(np.abs(fr - fp)))
    assert s1d[ipk - 1] > 0 and s1d[ipk] > 0             # energy moves to the forward face and the peak ...
    assert s1d[ipk + 3] < 0 and s1d[ipk + 4] < 0         # ... out of the band 1.3 - 1.5 fp (the DIA puts the negative lobe there) ...
    assert np.all(s1d[ipk + 9: ipk + 16] > 0)            # ... and into the tail
    # the transfer scales with the cube of the spectrum
    sl2, fld2 = np.zeros_like(fl), np.zeros_like(fl)
    fl2 = (2.0 * fl).copy()
    o.lib.ora_snonlin(_p(fl2), C.c_double(998.0), C.c_double(float(pr["WAVNUM"][0, 8])), _p(pr["WAVNUM"][0].copy()), _p(sl2), _p(fld2))
    assert np.allclose(sl2, 8.0 * sl, rtol=1e-12, atol=1e-300) and np.allclose(fld2, 4.0 * fld, rtol=1e-12, atol=1e-300)


def test_bottom_friction_formula():
    """SBOTTOM: S_bot = -2 * 0.038 / g * k / sinh(2 k d) * F (JONSWAP bottom friction, Hasselmann et al. 1973 with the
    empirical constant 0.038 m^2 s^-3), only on the NFRE_RED propagated frequencies, nothing in deep water."""
    cfg = Config(nang=12, nfre=36, nfre_red=29)
    t = Tables(cfg, np.float64)
    o = _oracle(cfg)
    d = 12.0
    pr = o.depthprpt(np.array([d]))
    fl = np.random.default_rng(0).uniform(0.1, 1.0, (12, 36))
    sl, fld = np.zeros_like(fl), np.zeros_like(fl)
    o.lib.ora_sbottom(_p(fl), _p(pr["WAVNUM"][0].copy()), C.c_double(d), _p(sl), _p(fld))
    k = pr["WAVNUM"][0]
    want = -2 * 0.038 / 9.806 * k / np.sinh(np.minimum(2 * k * d, 50.0))
    want[29:] = 0.0
    assert np.allclose(fld, np.broadcast_to(want, fl.shape), rtol=1e-12) and np.allclose(sl, fl * want[None, :], rtol=1e-12)
    o.lib.ora_sbottom(_p(fl), _p(pr["WAVNUM"][0].copy()), C.c_double(999.0), _p(sl), _p(fld))
    assert not sl.any() and not fld.any()


def test_saturation_dissipation_threshold():
    """SDISSIP_ARD (Ardhuin et al. 2010): no whitecapping where the directional saturation B(f, theta) stays below the
    threshold B_r = SDSBR (0.0009); above it the dissipation rate is negative and grows with the excess."""
    cfg = Config(nang=24, nfre=36, nfre_red=36)
    t = Tables(cfg, np.float64)
    o = _oracle(cfg)
    pr = o.depthprpt(np.array([998.0]))
    base = syn.jonswap_spectra(t.FR, t.TH, np.array([0.15]), np.array([0.5]), np.float64)[0]
    out = []
    for scale in (1e-4, 1.0, 3.0):
        fl = (scale * base).copy()
        sl, fld = np.zeros_like(fl), np.zeros_like(fl)
        o.lib.ora_sdissip_ard(_p(fl), _p(pr["WAVNUM"][0].copy()), _p(pr["XK2CG"][0].copy()), C.c_double(0.3), C.c_double(0.5), C.c_double(1.225),
                              _p(sl), _p(fld))
        out.append((sl.copy(), fld.copy()))
    assert not out[0][0].any() and not out[0][1].any()            # far below saturation: exactly nothing
    assert out[1][1].min() < 0 and out[1][1].max() <= 0            # a developed sea dissipates
    assert out[2][1].min() < 3.0 * out[1][1].min()                 # more than linearly in the energy


def test_drag_law_without_wave_stress():
    """TAUT_Z0 (Janssen 1991 quasi-linear theory): with no wave-induced stress the solution is the log profile over a
    Charnock sea with a viscous sublayer, u* = kappa U10 / ln(z_obs / (z0 + 0.11 nu / u*)) (z_obs = XNLEV = 10 m) with
    z0 = alpha u*^2 / g; the drag coefficient grows with wind speed and lies in the observed range."""
    cfg = Config(nang=12, nfre=36, nfre_red=25)
    t = Tables(cfg, np.float64)
    o = _oracle(cfg)
    cds = []
    for u10 in (5.0, 10.0, 20.0, 30.0):
        out = np.zeros(4)
        o.lib.ora_taut_z0(C.c_double(u10), C.c_double(0.3), C.c_double(0.0), C.c_double(0.3), C.c_double(0.0), C.c_double(1.0), _p(out))
        us, z0, z0b, ch = out
        z0vis = float(t.RNUM) / us
        assert abs(us - float(t.XKAPPA) * u10 / np.log(float(t.XNLEV) / (z0 + z0vis))) < 1e-6 * us   # log law over z0 + viscous sublayer
        assert abs(9.806 * z0 / us ** 2 - ch) < 1e-9 * ch                                            # CHRNCK is the Charnock parameter of z0
        assert 0.004 < ch < 0.04
        cds.append((us / u10) ** 2)
    # young-sea Charnock values (0.011-0.018) would give 1.2e-3 .. 2.3e-3; without wave stress ALPHA = 0.0065 gives less
    assert np.all(np.diff(cds) > 0) and 0.8e-3 < cds[0] < 1.3e-3 and 1.0e-3 < cds[1] < 1.4e-3 and 1.3e-3 < cds[2] < 2.0e-3
    # wave-induced stress roughens the sea: same wind, TAUW = 60 % of the total stress
    out0, out1 = np.zeros(4), np.zeros(4)
    o.lib.ora_taut_z0(C.c_double(15.0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_double(1.0), _p(out0))
    o.lib.ora_taut_z0(C.c_double(15.0), C.c_double(0.0), C.c_double(0.6 * out0[0] ** 2), C.c_double(0.0), C.c_double(0.0), C.c_double(1.0), _p(out1))
    assert out1[0] > out0[0] and out1[1] > out0[1]


# ---- the checks of tests/known_answers.py (written once, run here on the oracle and in test_gpu_known_answers.py on the device) ----------
import known_answers as KA  # noqa: E402


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_oracle_wind_input_growth_rate_and_momentum_balance(prec):
    run = KA.OracleRun(prec)
    KA.check_momentum_balance(run, prec, KA.check_wind_input_growth_rate(run, prec))


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_oracle_tail_and_stokes_drift(prec):
    KA.check_tail_and_stokes_drift(KA.OracleRun(prec), prec)


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_oracle_ctu_single_bin(prec):
    KA.check_ctu_single_bin(KA.OracleRun(prec), prec)


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_depth_props_of_the_product_and_of_the_oracle_agree(prec):
    """The two restatements of DEPTHPRPT / AKI (numpy in ecwam_amd/synthetic.py, scalar C in oracle/ora_tables.c) agree to a few units in
    the last place over the whole depth range -- including which Newton iteration AKI stops at (its convergence test has a tolerance of
    1e-4: one iteration more or less would show as 1e-5).  The parity harness feeds each side ITS OWN tables (harness.oracle_implsch)."""
    dt = H.np_dtype(prec)
    cfg = Config(nang=12, nfre=36, nfre_red=25)
    t = Tables(cfg, dt)
    o = _oracle(cfg, prec)
    rng = np.random.default_rng(11)
    depth = np.concatenate([10 ** rng.uniform(0.0, 3.0, 4000), [998.999, 1.0, 2.5, 7.0, 50.0, 49.999]])
    a, b = syn.depth_props(depth, t, dt), o.depthprpt(depth)
    eps = np.finfo(dt).eps
    for k in ("WAVNUM", "CINV", "CGROUP", "XK2CG", "OMOSNH2KD", "STOKFAC", "EMAXDPT"):
        x, y = a[k].astype(float), b[k].astype(float)
        ok = np.isfinite(y)
        err = np.abs(x - y)[ok] / np.maximum(np.abs(y[ok]), 1e-300)
        # OMOSNH2KD = omega / sinh(2 k d): an error e of the argument appears as 2 k d e in the result (2 k d up to 20 before the deep-water branch)
        assert np.array_equal(np.isfinite(x), ok) and err.max() <= (64 if k == "OMOSNH2KD" else 8) * eps, (k, float(err.max() / eps))
