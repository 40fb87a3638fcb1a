"""GPU tests of the one-kernel WAMINTGR step (round 6; run with -m gpu on an MI355X): ecwam_hip_propags2_implsch = PROPAGS2
(propags2.F90:99-121, IREFRA = 0) inside the tile load of the source-term kernel, against

* the two-kernel path through the same C ABI (ecwam_hip_propags2_otf + ecwam_hip_newwind + ecwam_hip_implsch): BIT-IDENTICAL spectra, forcing
  and integrated fields, MIJ and XLLWS, step after step -- the advecting load uses the arithmetic of the stencil kernel (csrc/ctu.h), the
  rest of the kernel is the same source;
* the oracle stepping the same state (wamintgr.F90:94-146: PROPAG_WAM, NEWWIND, IMPLSCH), under the single-precision gates every other
  comparison uses (tests/harness.py);
* itself on a decomposed grid (three contiguous sea-point bands with halo rows, decomp.py / mpdecomp.F90:58-100): bit-identical to the
  single domain.
"""
import numpy as np
import pytest

import harness as H
from ecwam_amd.tables import Config

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def api():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import api as _api

    return _api


def _pair(cfg, g, seed, prec="sp", **kw):
    from ecwam_amd.wamintgr import Wamintgr

    ms = []
    for _ in range(2):
        m = Wamintgr(cfg, g, prec, **kw)
        m.init_synthetic(seed=seed)
        m.ff_next = m.ff.clone()
        m.ff_next[:, 3] *= 1.03      # a new wind speed: NEWWIND hands it over before the source terms
        ms.append(m)
    return ms


def _same_state(a, b):
    n = a.n
    for name in ("fl1", "ff", "intf", "mij", "xllws"):
        x, y = getattr(a, name)[:n], getattr(b, name)[:n]
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} of {x.numel()} elements differ"


@pytest.mark.parametrize("prec", ["sp", "dp"])
@pytest.mark.parametrize("flags", [dict(), dict(llgcbz0=True, llnormagam=True), dict(lciwa3=True, lciscal=True, lciwa1=True)])
@pytest.mark.parametrize("nfre_red,ngrid", [(36, 24), (29, 17)])
def test_one_kernel_step_is_bit_identical_to_the_two_kernels(api, flags, nfre_red, ngrid, prec):
    """Three steps on a grid with land (land slot, short last wave: the sea-point count is not a multiple of three at either size), all
    frequencies advected or the last seven carried over (NFRE_RED = 29 cuts a 16-byte vector)."""
    from ecwam_amd import grid as G

    cfg = Config(nang=36, nfre=36, nfre_red=nfre_red, idelt=450, idelpro=450, **flags)
    g = G.build_grid(ngrid, mask="continents")
    two, one = _pair(cfg, g, seed=21, prec=prec)
    if flags.get("lciwa3"):      # sea-ice attenuation (common builds: the rates depend on the frequency only): partial ice cover on a third of the points
        rng = np.random.default_rng(8)
        for m in (two, one):
            ice = torch.from_numpy(np.where(rng.uniform(size=m.n) < 0.33, rng.uniform(0.05, 0.95, m.n), 0.0)).to(m.dev, m.dtype)
            thick = torch.from_numpy(rng.uniform(0.1, 3.0, m.n)).to(m.dev, m.dtype)
            for ff in (m.ff, m.ff_next):
                ff[:, 2] = ice
                ff[:, 13] = thick
            rng = np.random.default_rng(8)
    assert one.fused_available()
    assert two.build_weights() == 0 and one.build_weights() == 0
    for _ in range(3):
        two.step()
        one.step(fused=True)
        torch.cuda.synchronize()
        _same_state(two, one)
    assert float(one.fl1[: one.n].abs().max()) > 0 and bool(torch.isfinite(one.fl1).all())
    two.ctx.close(); one.ctx.close()


@pytest.mark.parametrize("prec", ["sp", "dp"])
@pytest.mark.parametrize("nfre_red,lfm", [(36, 5), (29, 4)])
def test_one_kernel_step_with_fast_wave_sub_steps_is_bit_identical(api, prec, nfre_red, lfm):
    """The native O1280 structure (propag_wam.F90:247-313): fast waves M <= IFRELFMAX in two sub-steps of half the time step on compact rows.
    The one-kernel step runs the first sub-step as PROPAGS2 compact -> compact and takes the last one -- together with the slow waves' step --
    into the tile load (ADV = 3: compact input rows, per-frequency time step), and leaves the new fast waves in the compact rows the next
    step starts from; three steps, so that the compact rows are handed from step to step.  IFRELFMAX = 5 cuts a 16-byte vector."""
    from ecwam_amd import grid as G

    cfg = Config(nang=36, nfre=36, nfre_red=nfre_red, idelt=450, idelpro=450)
    g = G.build_grid(20, mask="continents")
    two, one = _pair(cfg, g, seed=33, prec=prec, ifrelfmax=lfm, delpro_lf=225.0)
    assert one.fused_available()
    assert two.build_weights() == 0 and one.build_weights() == 0
    for _ in range(3):
        two.step()
        one.step(fused=True)
        torch.cuda.synchronize()
        _same_state(two, one)
        assert one.gfast_valid and torch.equal(one.g1[: one.n], two.g1[: two.n])
    # two advection steps per source step (the O1280 cycle): PROPAG_WAM, then the one-kernel step
    for m, fused in ((two, False), (one, True)):
        m.propag()
        m.step(fused=fused)
    torch.cuda.synchronize()
    _same_state(two, one)
    two.ctx.close(); one.ctx.close()


@pytest.mark.parametrize("prec,lfm", [("sp", 0), ("dp", 0), ("sp", 5), ("dp", 4)])
def test_one_kernel_step_with_sub_grid_obstructions_is_bit_identical(api, prec, lfm):
    """LSUBGRID (the reference's default on real bathymetry, mpuserin.F90:704): the transmission coefficients OBSLAT / OBSLON / OBSCOR scale the
    space weights of the neighbours (ctuw.F90:703-733).  The tile load fetches the three coefficient vectors of a step's neighbours with its
    gathers (ADV = 5; with fast-wave sub-steps ADV = 7) and applies them as k_propags2_otf does: three steps, the same bits; and the table
    acts (the result differs from the run without it)."""
    from ecwam_amd import grid as G, synthetic as syn

    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    g = G.build_grid(20, mask="continents")
    kw = dict(ifrelfmax=lfm, delpro_lf=225.0) if lfm else {}
    two, one = _pair(cfg, g, seed=17, prec=prec, **kw)
    plain, _ = _pair(cfg, g, seed=17, prec=prec, **kw)
    _.ctx.close()
    obs = syn.obstructions(g, cfg.nfre, seed=5)
    for m in (two, one):
        m.set_obstructions(obs)
        assert m.build_weights() == 0
    assert plain.build_weights() == 0 and one.fused_available()
    for _i in range(3):
        two.step()
        one.step(fused=True)
        plain.step(fused=True)
        torch.cuda.synchronize()
        _same_state(two, one)
    assert not torch.equal(one.fl1[: one.n], plain.fl1[: plain.n])
    for m in (two, one, plain):
        m.ctx.close()


@pytest.mark.parametrize("prec", ["sp", "dp"])
@pytest.mark.parametrize("nang,nfre_red,lfm,subgrid", [(24, 29, 0, False), (24, 29, 5, True), (12, 25, 0, False), (12, 25, 4, True), (48, 36, 0, False),
                                                       (48, 36, 0, True)])
def test_one_kernel_step_at_the_other_direction_counts(api, prec, nang, nfre_red, lfm, subgrid):
    """The spectral grids of the reference's registered configurations (24 x 29 and 12 x 25 of NFRE = 36: tests/CMakeLists.txt:11-46,
    etopo1_oper_an_fc_O320.yml:2-3) and 48 directions: 5 / 10 / 2 points per wavefront, other round and tail counts of the advecting load
    (24 directions: 3 rounds + 24 left-over chunks per point = two tail steps; 12: 1 round + 44 = seven), with fast-wave sub-steps and
    obstructions where a build holds them (48 directions: no fast-wave form).  Three steps, bit-identical to the two kernels."""
    from ecwam_amd import grid as G, synthetic as syn

    cfg = Config(nang=nang, nfre=36, nfre_red=nfre_red, idelt=450, idelpro=450)
    g = G.build_grid(17, mask="continents")
    kw = dict(ifrelfmax=lfm, delpro_lf=225.0) if lfm else {}
    two, one = _pair(cfg, g, seed=29, prec=prec, **kw)
    if prec == "dp":
        # double precision has one-kernel builds at 36 directions only (implsch4a.hip: the others were miscompiled at -O3 and are not
        # shipped): the model must say so and step(fused=True) must fall back to the two kernels -- still the same bits
        assert not one.fused_available()
    if subgrid:
        obs = syn.obstructions(g, cfg.nfre, seed=5)
        obs[:, :, nfre_red:] = 1.0
        for m in (two, one):
            m.set_obstructions(obs)
    assert two.build_weights() == 0 and one.build_weights() == 0 and (prec == "dp" or one.fused_available())
    for _i in range(3):
        two.step()
        one.step(fused=True)
        torch.cuda.synchronize()
        _same_state(two, one)
    assert float(one.fl1[: one.n].abs().max()) > 0 and bool(torch.isfinite(one.fl1).all())
    two.ctx.close(); one.ctx.close()


def test_one_kernel_step_natural_order_and_row_blocks(api):
    """The workgroups in the XCD-aware order (flags bit 0) and the rows passed in three unequal blocks give the same bits as one call."""
    from ecwam_amd import grid as G

    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    g = G.build_grid(20, mask="continents")
    a, b = _pair(cfg, g, seed=5)
    assert a.build_weights() == 0 and b.build_weights() == 0
    a.step_fused()
    b.newwind()
    n = b.n
    for k0, k1, fl in ((0, 7, 1), (7, n // 2 + 1, 0), (n // 2 + 1, n, 1)):
        b.ctx.propags2_implsch(b.fl1, b.fl3, b.gd, b.cgroup_ext, float(cfg.idelpro), k0, k1, b.wvprpt, b.ff, b.intf, b.mij, b.xllws, 1, cfg.nfre_red,
                               flags=fl)
    b.fl1, b.fl3 = b.fl3, b.fl1
    torch.cuda.synchronize()
    _same_state(a, b)
    a.ctx.close(); b.ctx.close()


def test_one_kernel_step_matches_oracle(api):
    """Two one-kernel steps against the oracle's PROPAGS2 + IMPLSCH on the same state (single precision gates of tests/harness.py)."""
    from ecwam_amd import grid as G
    from ecwam_amd.wamintgr import Wamintgr
    from oracle.oracle import Oracle

    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    g = G.build_grid(16, mask="continents")
    m = Wamintgr(cfg, g, "sp")
    m.init_synthetic(seed=4)
    assert m.fused_available() and m.build_weights() == 0
    o = Oracle(cfg, "sp")
    n = g.nsea
    fl = m.fl1.cpu().numpy().copy()
    wv = m.wvprpt.cpu().numpy()
    ff = m.ff.cpu().numpy()[:, :14].copy()
    env = m.ff.cpu().numpy()[:, 14:16].copy()
    intf = np.zeros((n, 15), np.float32)
    wref = o.ctu_weights(g, m.cgroup_ext.cpu().numpy(), float(cfg.idelpro))
    for _ in range(2):
        m.step(fused=True)
        f3 = o.propags2(g, fl, wref)
        r = o.implsch(f3[:n], wv[:, 0], wv[:, 1], wv[:, 2], wv[:, 3], wv[:, 4], env, ff, intf)
        fl[:n], ff, intf = r["FL1"], r["FF"], r["INTF"]
    torch.cuda.synchronize()
    got = {"FL1": m.fl1[:n].cpu().numpy(), "MIJ": m.mij.cpu().numpy(), "FF": m.ff.cpu().numpy()[:, :14], "INTF": m.intf.cpu().numpy()[:, :15],
           "XLLWS": m.xllws.cpu().numpy()}
    st = H.compare_implsch(r, got, m.t)
    H.assert_sp_gates(st, n)
    m.ctx.close()


def test_one_kernel_step_on_a_decomposed_grid_is_bit_identical(api):
    """Three bands with halo rows filled by hand from the neighbours' owned rows (what MPEXCHNG delivers), interior and edge rows in
    separate calls as Wamintgr.step_fused issues them: two steps, bit-identical to the single domain."""
    from ecwam_amd import grid as G
    from ecwam_amd.wamintgr import Wamintgr

    cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=450, idelpro=450)
    g = G.build_grid(20, mask="continents")
    ref = Wamintgr(cfg, g, "sp")
    ref.init_synthetic(seed=11)
    nr = 3
    parts = []
    for r in range(nr):
        m = Wamintgr(cfg, g, "sp", rank=r, nranks=nr)
        m.init_synthetic(seed=11)
        parts.append(m)

    class ByHand:      # HaloExchange stand-in: the halo rows straight from the other bands' owned rows
        def __init__(self, m):
            self.m = m

        def start(self, fl):
            glob = torch.cat([q.fl1[: q.n] for q in parts])
            hg = torch.from_numpy(np.asarray(self.m.dom.halo_global, dtype=np.int64)).to(glob.device)
            fl[self.m.n: self.m.n + self.m.dom.nh] = glob[hg]
            return []

        def finish(self, reqs):
            pass

    for m in parts:
        m.halo = ByHand(m)
        assert m.fused_available() and m.build_weights() == 0
    assert ref.build_weights() == 0
    for _ in range(2):
        ref.step(fused=True)
        # every band reads the OLD owned rows of its neighbours: fill all halos before anybody swaps its buffers
        for m in parts:
            m.halo.start(m.fl1)
        keep = [m.halo for m in parts]
        for m in parts:
            m.halo = type("Done", (), {"start": staticmethod(lambda fl: []), "finish": staticmethod(lambda r: None)})()
        for m in parts:
            m.step(fused=True)
        for m, h in zip(parts, keep):
            m.halo = h
    torch.cuda.synchronize()
    got = torch.cat([m.fl1[: m.n] for m in parts])
    assert torch.equal(got, ref.fl1[: ref.n])
    assert torch.equal(torch.cat([m.mij for m in parts]), ref.mij)
    for m in parts + [ref]:
        m.ctx.close()


def test_strict_build_of_the_weights_is_bit_identical_to_the_stored_weight_scheme(api):
    """The library built with -DECWAM_HIP_CTU_STRICT=1 (build variant "ctustrict": the on-the-fly weights in ctuw.F90's order of operations,
    contraction off) in a child process: on-the-fly and stored weights give the same bits (tests/harness.py: CTU_STRICT), and the one-kernel
    step still equals the two kernels."""
    import os
    import subprocess
    import sys

    from ecwam_amd import build

    lib = build.lib_path("ctustrict")
    if not os.path.exists(lib):
        build.build(variant="ctustrict")
    env = dict(os.environ, ECWAM_HIP_LIB=lib, ECWAM_TEST_CTU_STRICT="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(here, "test_gpu_parity.py"), os.path.join(here, "test_gpu_refraction.py"),
                        os.path.join(here, "test_gpu_fused.py"), "-k",
                        "ctuw_and_propags2_parity or subgrid_obstructions or one_kernel_step_is_bit_identical or zero_currents_reduce"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
