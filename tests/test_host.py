"""CPU tests: the C ABI library loads and exports every declared symbol, host-side grid / decomposition logic,
the N>1 halo-exchange path over gloo (world_size 2), and oracle invariants."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import harness as H
from ecwam_amd import decomp, grid as G
from ecwam_amd.tables import Config, Tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from ecwam_amd import build, lib

    build.build()
    h = lib.load()
    hdr = open(os.path.join(ROOT, "include", "ecwam_hip.h")).read()
    declared = set(re.findall(r"\b(ecwam_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for name in declared:
        assert getattr(h, name) is not None
    assert h.ecwam_hip_abi_version() == int(re.search(r"#define ECWAM_HIP_ABI_VERSION (\d+)", hdr).group(1)) == 6
    # the parameter struct seen from Python has the size the C compiler gives it
    src = '#include "ecwam_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu", sizeof(ecwam_hip_params), sizeof(ecwam_hip_tables));return 0;}'
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(), "abi_sizes")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b = (int(x) for x in subprocess.run([exe], capture_output=True, check=True).stdout.split())
    import ctypes
    assert a == ctypes.sizeof(lib.Params) and b == ctypes.sizeof(lib.TablePtrs)


def test_create_refuses_unsupported_configurations_without_a_gpu():
    """Error convention of the boundary: int status + ecwam_hip_last_error(); the configuration checks of
    ecwam_hip_create run before any HIP call, so they are testable here."""
    import ctypes as C
    from ecwam_amd import lib

    h = lib.load()
    t = Tables(Config(nang=12, nfre=36, nfre_red=25), np.float32)
    tp, keep = lib.make_tables(t)
    cases = [(dict(lciwa1=1), "SDICE1"), (dict(iphys=2), "IPHYS"), (dict(irefra=4), "IREFRA"), (dict(isnonlin=3), "ISNONLIN"), (dict(icode=4), "ICODE"),
             (dict(nang=3), "NANG"), (dict(nfre_red=40), "NFRE_RED")]
    for changes, word in cases:
        p = lib.make_params(t)
        for k, v in changes.items():
            setattr(p, k, v)
        tp.cideac = None if "lciwa1" in changes else keep[-1].ctypes.data_as(C.c_void_p)   # SDICE1 without its table is refused
        ctx = C.c_void_p()
        rc = h.ecwam_hip_create(C.byref(p), C.byref(tp), 4, 0, C.byref(ctx))
        assert rc != 0 and not ctx.value
        msg = h.ecwam_hip_last_error().decode()
        assert word in msg, (word, msg)
    p = lib.make_params(t)
    assert h.ecwam_hip_create(C.byref(p), C.byref(tp), 2, 0, C.byref(C.c_void_p())) != 0      # real_bytes must be 4 or 8
    assert h.ecwam_hip_create(None, C.byref(tp), 4, 0, C.byref(C.c_void_p())) != 0


def test_restart_spectra_file_layout(tmp_path, monkeypatch):
    """writefl.F90:110-118 layout: one unformatted sequential record, FL(IJ,K,M) with IJ fastest; framed so that a Fortran
    reader (checked with scipy.io.FortranFile, which implements the same 4-byte markers) sees it; sub-record splitting of
    huge records exercised with a lowered limit."""
    from scipy.io import FortranFile
    from ecwam_amd import restart

    rng = np.random.default_rng(0)
    n, K, M = 37, 12, 25
    fl = rng.uniform(0, 1, (n, K, M)).astype(np.float32)
    f = str(tmp_path / "BLS_test")
    restart.write_fl(f, fl)
    restart.write_fl(f, 2 * fl, append=True)
    with FortranFile(f, "r") as ff:
        rec = ff.read_reals(dtype=np.float32)
    assert rec.size == n * K * M
    assert np.array_equal(rec.reshape(M, K, n)[3, 5, :], fl[:, 5, 3])          # IJ runs fastest in the record
    assert np.array_equal(restart.read_fl(f, n, K, M, np.float32), fl)
    assert np.array_equal(restart.read_fl(f, n, K, M, np.float32, record=1), 2 * fl)
    with pytest.raises(ValueError):
        restart.read_fl(f, n + 1, K, M, np.float32)
    monkeypatch.setattr(restart, "_MAX_SUB", 1000)                                # force gfortran-style sub-records
    g = str(tmp_path / "BLS_big")
    fd = fl.astype(np.float64)
    restart.write_fl(g, fd)
    assert np.array_equal(restart.read_fl(g, n, K, M, np.float64), fd)
    raw = open(g, "rb").read()
    assert int(np.frombuffer(raw[:4], "<i4")[0]) == -1000 and int(np.frombuffer(raw[-4:], "<i4")[0]) < 0


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ecwam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".F90")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("test oracle", "").lower() or f in ("lib.py",), (dirpath, f)


def test_product_never_touches_the_tests_second_implementation():
    """k_implsch2 (tests/csrc/implsch_v2.h) left the product in round 5: nothing under ecwam_amd/ includes, builds or loads it, and the
    product library exports no entry point that selects a kernel generation."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ecwam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".F90")):
                txt = open(os.path.join(dirpath, f)).read()
                for word in ("v2lib", "libecwam_v2", "k_implsch2<", "launch_implsch<", 'include "implsch_v2.h"', "implsch_wave_v2"):
                    assert word not in txt, (dirpath, f, word)
    hdr = open(os.path.join(ROOT, "include", "ecwam_hip.h")).read()
    assert "int ecwam_hip_set_implsch_generation" not in hdr


def test_grid_counts_and_neighbours():
    for n in (16, 24, 48):
        g = G.build_grid(n)
        assert g.nsea == G.nsea_aqua(n)
    g = G.build_grid(24, mask="continents")
    nl = g.nland
    assert g.klon.min() >= 0 and g.klon.max() <= nl and g.klat.max() <= nl and g.kcor.max() <= nl
    sea = g.klon[:, 1] < nl                         # east neighbour of a point has that point as west neighbour
    assert np.array_equal(g.klon[g.klon[sea, 1], 0], np.flatnonzero(sea))
    for ic, dk in ((0, -1), (1, 1)):                # latitude neighbours live in the adjacent row
        ok = g.klat[:, ic, 0] < nl
        assert np.all(g.kxlt[g.klat[ok, ic, 0]] == g.kxlt[ok] + dk)
        ok = g.kcor[:, 0 if dk > 0 else 1, 0] < nl
        assert np.all(g.kxlt[g.kcor[ok, 0 if dk > 0 else 1, 0]] == g.kxlt[ok] + dk)
    assert 0.0 < g.wlat.min() and g.wlat.max() <= 1.0 and 0.0 <= g.wcor.min() and g.wcor.max() <= 1.0


@pytest.mark.parametrize("name", ["O48", "O320", "O640"])
def test_grid_matches_reference_grid_description(name):
    """Golden vectors produced by the reference's own grid script (tools/make_golden_grid.py ran
    share/ecwam/scripts/ecwam_grids.py here): resolution, first latitude, east-most longitude, ny and every row length of
    the octahedral grids of the BASELINE configurations against ecwam_amd.grid."""
    f = os.path.join(ROOT, "tests", "golden", f"grid_description_{name}.txt")
    v = open(f).read().split()
    n, north, south, west, east, iper, irgg, ny = int(v[0]), float(v[1]), float(v[2]), float(v[3]), float(v[4]), int(v[5]), int(v[6]), int(v[7])
    rows = np.array([int(x) for x in v[8:8 + ny]])
    g = G.build_grid(n)
    assert g.ngy == ny == 2 * n and iper == 1 and irgg == 1 and west == 0.0
    assert np.array_equal(np.asarray(g.nlonrgg), rows)
    assert abs(float(g.amosop) - south) < 1e-12 and abs(float(g.xdella) - (north - south) / (ny - 1)) < 1e-12
    assert abs(360.0 - 360.0 / rows.max() - east) < 1e-9
    assert g.nsea == G.nsea_aqua(n)


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_decomposition_consistency(nranks):
    g = G.build_grid(24, mask="continents")
    doms = [decomp.local_domain(g, r, nranks) for r in range(nranks)]
    assert sum(d.n for d in doms) == g.nsea and max(d.n for d in doms) - min(d.n for d in doms) <= 1
    for d in doms:
        ext = np.concatenate([d.ext_global(), [g.nland]])
        for loc, glo in ((d.klon, g.klon), (d.klat, g.klat), (d.kcor, g.kcor)):
            assert np.array_equal(ext[loc], glo[d.lo:d.hi])
        for p, (dst0, cnt) in d.recv.items():       # what I expect from p is exactly what p plans to send me
            want = d.halo_global[dst0 - d.n: dst0 - d.n + cnt]
            assert np.array_equal(doms[p].send[d.rank] + doms[p].lo, want)
        assert set(d.recv) == {p for p in range(nranks) if d.rank in doms[p].send}


def test_oracle_decomposed_advection_equals_global():
    """Advecting each rank's local domain (with halo rows filled from the neighbours) reproduces the global result
    bit for bit: the stencil is a pure gather (decomposition independence)."""
    from oracle.oracle import Oracle

    cfg = Config(nang=12, nfre=36, nfre_red=25)
    g = G.build_grid(16, mask="continents")
    o = Oracle(cfg, "dp")
    rng = np.random.default_rng(0)
    cg = np.zeros((g.nsea + 1, 36))
    cg[:] = rng.uniform(3, 12, (g.nsea + 1, 36))
    f1 = np.zeros((g.nsea + 1, 12, 36))
    f1[: g.nsea] = rng.uniform(0, 1, (g.nsea, 12, 36))
    w = o.ctu_weights(g, cg, 900.0)
    f3 = o.propags2(g, f1, w)

    class LG:  # a Grid-like view of one local domain
        pass

    for r in range(3):
        d = decomp.local_domain(g, r, 3)
        ext = np.concatenate([d.ext_global(), [g.nland]])
        lg = LG()
        lg.nsea, lg.nland, lg.ngy, lg.xdella = d.n, d.nland, g.ngy, g.xdella
        lg.kxlt, lg.klon, lg.klat, lg.kcor = d.kxlt, d.klon, d.klat, d.kcor
        lg.wlat, lg.wcor = g.wlat[d.lo:d.hi], g.wcor[d.lo:d.hi]
        lg.cosph, lg.sinph, lg.zdello, lg.cosphm1_ext = g.cosph, g.sinph, g.zdello, d.cosphm1_ext
        wl = o.ctu_weights(lg, cg[ext], 900.0)
        f3l = o.propags2(lg, f1[ext], wl)
        assert np.array_equal(f3l[: d.n], f3[d.lo:d.hi])


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_oracle_propag_wam_is_the_reference_call_sequence(prec):
    """ora_propag_wam (propag_wam.F90:247-313) against the same sequence written out call by call with an independent numpy
    stencil of propags2.F90:107-116 fed with the oracle's weight arrays: the two time steps meet in one weight set
    (ctuwupdt.F90:220-256), the fast waves are advected NSTEP_LF times with DELPRO_LF, the slow waves once with IDELPRO."""
    from oracle.oracle import Oracle

    cfg = Config(nang=12, nfre=36, nfre_red=25, idelpro=900)
    g = G.build_grid(12, mask="continents")
    o = Oracle(cfg, prec)
    dt = np.float32 if prec == "sp" else np.float64
    t = Tables(cfg, dt)
    rng = np.random.default_rng(3)
    n, NANG, NR, lf, dlf = g.nsea, 12, 25, 6, 300.0
    cg = rng.uniform(3, 12, (n + 1, 36)).astype(dt)
    f1 = np.zeros((n + 1, NANG, 36), dt)
    f1[:n] = rng.uniform(0, 1, (n, NANG, 36))
    w = o.ctu_weights_wam(g, cg, cfg.idelpro, lf, dlf)
    wf = o.ctu_weights(g, cg, dlf)            # every frequency with the fast-wave step
    ws = o.ctu_weights(g, cg, 900.0)          # every frequency with the full step
    for k in ("SUMWN", "WLONN", "WLATN", "WCORN", "WKPMN"):
        assert np.array_equal(w[k][:, :, :lf], wf[k][:, :, :lf]) and np.array_equal(w[k][:, :, lf:], ws[k][:, :, lf:])
    assert w["NFAIL"] == 0

    jxo, jyo, kcr = np.asarray(t.JXO)[:, 0] - 1, np.asarray(t.JYO)[:, 0] - 1, np.asarray(t.KCR)[:, 0] - 1
    K = np.arange(NANG)
    km, kp = (K - 1) % NANG, (K + 1) % NANG

    def stencil(f, m0, m1):
        """F3(IJ,K,M) of propags2.F90:107-116 for M in [m0, m1), terms in the reference's order."""
        out = np.zeros((n, NANG, m1 - m0), dt)
        for k in range(NANG):
            sl = (slice(0, n), k, slice(m0, m1))
            fo = f[:n, k, m0:m1]
            acc = (dt(1.0) - w["SUMWN"][sl]) * fo
            acc = acc + w["WLONN"][sl + (jxo[k],)] * f[g.klon[:, jxo[k]], k, m0:m1]
            for icl in range(2):
                acc = acc + w["WLATN"][sl + (jyo[k], icl)] * f[g.klat[:, jyo[k], icl], k, m0:m1]
            for icl in range(2):
                acc = acc + w["WCORN"][sl + (0, icl)] * f[g.kcor[:, kcr[k], icl], k, m0:m1]
            acc = acc + w["WKPMN"][sl + (0,)] * f[:n, km[k], m0:m1]
            acc = acc + w["WKPMN"][sl + (2,)] * f[:n, kp[k], m0:m1]
            out[:, k] = acc
        return out

    f3 = f1.copy()
    f3[:n, :, :NR] = stencil(f1, 0, NR)
    ext = f1.copy()
    for _ in range(2, 4):                      # NSTEP_LF = NINT(900 / 300) = 3: sub-steps 2 and 3
        ext[:n, :, :lf] = f3[:n, :, :lf]
        f3[:n, :, :lf] = stencil(ext, 0, lf)
    got, nstep = o.propag_wam(g, f1, w, cfg.idelpro, lf, dlf)
    assert nstep == 3
    assert np.array_equal(got[:n, :, NR:], f1[:n, :, NR:]) and not got[n].any()
    eps = np.finfo(dt).eps
    assert np.abs(got[:n, :, :NR].astype(float) - f3[:n, :, :NR].astype(float)).max() <= 4 * eps      # numpy may fuse nothing: same order
    # without sub-steps the composed call is PROPAGS2 itself
    one, nstep = o.propag_wam(g, f1, ws, cfg.idelpro, 0, None)
    ref = o.propags2(g, f1, ws)
    assert nstep == 1 and np.array_equal(one[:n, :, :NR], ref[:n, :, :NR])


def test_oracle_physics_invariants():
    """Properties the domain offers: limiter bounds, noise floor, ice mask, f^-5 tail continuity (implsch.F90:384-391,
    imphftail.F90:73-87, setice.F90:67-86)."""
    from oracle.oracle import Oracle

    cfg = Config(nang=24, nfre=36, nfre_red=29)
    case = H.make_point_case(400, cfg, "dp", spectra="mixed", seed=11)
    t = case["tables"]
    r = H.oracle_implsch(case, Oracle(cfg, "dp"))
    fl, mij = r["FL1"], r["MIJ"]
    assert np.isfinite(fl).all() and fl.min() >= 0
    # the limiter's cap (implsch.F90:386-391) holds on the prognostic rows; above MIJ the diagnostic tail of row MIJ goes in afterwards
    # (imphftail.F90:77-88) and may pass FLMAX(M) by its finite-depth factor (7.8e-5 seen with another seed, ECWAM_TEST_SEED_OFFSET)
    prog = np.arange(36)[None, None, :] < mij[:, None, None]
    assert np.all((fl <= np.asarray(t.FLMAX)[None, None, :] * (1 + 1e-12) + 1e-30) | ~prog)
    assert np.all(fl <= np.asarray(t.FLMAX)[None, None, :] * (1 + 1e-3))
    ice = case["FF"][:, 2] > float(t.CITHRSH)
    assert ice.any() and np.all(fl[ice].max(axis=(1, 2)) < 1e-4)
    assert np.all((mij >= 1) & (mij <= 36))
    free = ~ice
    for ij in np.flatnonzero(free)[:50]:            # above MIJ the spectrum follows F(MIJ) * k-dependent factor (or the floor)
        m = mij[ij]
        if m < 36:
            pr = case["props"]
            fac = (pr["XK2CG"][ij, m - 1] * pr["WAVNUM"][ij, m - 1]) / (pr["XK2CG"][ij, m:] * pr["WAVNUM"][ij, m:])
            tail = fl[ij, :, m - 1][:, None] * fac[None, :]
            assert np.all((np.abs(fl[ij, :, m:] - tail) <= 1e-12 * tail + 1e-30) | (fl[ij, :, m:] >= tail))


def test_oracle_optional_branches_invariants():
    """The optional branches the device also implements, checked on the oracle alone (CPU): flag set B runs and differs
    from A; sea-ice attenuation only removes energy and only under ice; the NEMO accumulators add up linearly; the OUTBS
    subset is consistent with SEMEAN-type integrals; strip work order is a permutation with better neighbour locality."""
    from oracle.oracle import Oracle

    base = dict(nang=24, nfre=36, nfre_red=29)
    n = 300
    cfg_a = Config(**base)
    case = H.make_point_case(n, cfg_a, "dp", spectra="mixed", seed=5)
    rng = np.random.default_rng(2)
    case["FF"][:, 2] = np.where(rng.uniform(size=n) < 0.5, rng.uniform(0.05, 0.29, n), 0.0)   # partial ice below the mask threshold
    case["FF"][:, 13] = rng.uniform(0.5, 3.0, n)
    ra = H.oracle_implsch(case, Oracle(cfg_a, "dp"))
    # flag set B
    cfg_b = Config(llgcbz0=True, llnormagam=True, **base)
    cb = dict(case); cb["cfg"] = cfg_b; cb["tables"] = Tables(cfg_b, np.float64)
    rb = H.oracle_implsch(cb, Oracle(cfg_b, "dp"))
    assert np.isfinite(rb["FL1"]).all() and np.isfinite(rb["FF"]).all()
    assert np.max(np.abs(rb["FF"][:, 7] - ra["FF"][:, 7]) / ra["FF"][:, 7]) > 1e-3          # UFRIC from the other roughness model
    # sea-ice attenuation
    dfim = np.asarray(case["tables"].DFIM, dtype=float)
    ice = case["FF"][:, 2] > 0
    ea = (ra["FL1"].sum(1) * dfim).sum(1)
    for flags, monotone in ((dict(lciwa3=True), True), (dict(lciwa3=True, lciscal=True), False)):
        cfg_i = Config(**flags, **base)
        ci = dict(case); ci["cfg"] = cfg_i; ci["tables"] = Tables(cfg_i, np.float64)
        ri = H.oracle_implsch(ci, Oracle(cfg_i, "dp"))
        ei = (ri["FL1"].sum(1) * dfim).sum(1)
        assert np.array_equal(ri["FL1"][~ice], ra["FL1"][~ice])                               # no ice: untouched
        assert np.any(np.abs(ei[ice] - ea[ice]) > 1e-6 * ea[ice])
        if monotone:   # a pure attenuation term only removes energy (LCISCAL also scales the dissipation down)
            assert np.all(ei[ice] <= ea[ice] * (1 + 1e-12)) and np.any(ei[ice] < ea[ice] * 0.999)
    # NEMO accumulators
    cfg_n = Config(lwnemocou=True, **base)
    o = Oracle(cfg_n, "dp")
    pr = case["props"]
    args = (pr["WAVNUM"], pr["CGROUP"], pr["CINV"], pr["XK2CG"], pr["STOKFAC"], case["ENV"])
    w0 = np.zeros((n, 13))
    r1 = o.implsch(case["FL1"], *args, case["FF"], case["INTF"], w2n=w0)
    r1b = o.implsch(case["FL1"], *args, case["FF"], case["INTF"], w2n=r1["W2N"])
    assert np.allclose(r1b["W2N"][:, [7, 8, 11, 12]], 2 * r1["W2N"][:, [7, 8, 11, 12]], rtol=1e-13, atol=0)
    assert np.array_equal(r1b["W2N"][:, 3:7], r1["W2N"][:, 3:7]) and np.all(r1["W2N"][:, 5] > 0)
    assert np.allclose(r1["W2N"][:, 0], r1["INTF"][:, 2]) and np.allclose(r1["W2N"][:, 1], r1["INTF"][:, 3])
    # OUTBS subset
    ob = Oracle(cfg_a, "dp").outbs(case["FL1"])
    assert np.allclose(ob[:, 0], 4 * np.sqrt(ob[:, 3])) and np.all((ob[:, 1] >= 0) & (ob[:, 1] < 360)) and np.all(ob[:, 2] > 0)
    # strip order
    g = G.build_grid(24, mask="continents")
    d = decomp.local_domain(g, 0, 1)
    order = decomp.strip_order(g, d, 32)
    assert sorted(order.tolist()) == list(range(d.n))
    # 2-D tile order: every point once, padded to whole tiles, a tile spans at most four latitude rows
    t2 = decomp.tile2d_order(g, d)
    assert t2.shape[0] % 16 == 0 and sorted(t2[t2 >= 0].tolist()) == list(range(d.n))
    ky = np.asarray(g.kxlt)[d.lo:d.hi]
    for tile in t2.reshape(-1, 16):
        rows = ky[tile[tile >= 0]]
        assert rows.size and rows.max() - rows.min() <= 3


GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ecwam_amd import decomp, grid as G
from ecwam_amd.wamintgr import HaloExchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
g = G.build_grid(16, mask="continents")
d = decomp.local_domain(g, rank, world)
rng = np.random.default_rng(42)
h = HaloExchange(d, torch.device("cpu"))
ok = d.nh > 0
for shape in ((4, 6), (4, 2), (4, 6)):                 # full rows, compact fast-wave rows, full rows again (buffers per shape)
    glob = rng.uniform(0, 1, (g.nsea,) + shape)        # same on every rank
    fl = torch.zeros((d.nrows,) + shape, dtype=torch.float64)
    fl[: d.n] = torch.from_numpy(glob[d.lo:d.hi])
    reqs = h.start(fl)                                 # posted ...
    a, b = d.interior()
    interior_sum = float(fl[a:b].sum())                # ... the interior rows are usable meanwhile ...
    h.finish(reqs)                                     # ... the halo rows after the wait
    ok = ok and np.array_equal(fl[d.n:d.n + d.nh].numpy(), glob[d.halo_global]) and float(fl[d.nland].abs().sum()) == 0.0
    ok = ok and abs(interior_sum - glob[d.lo + a:d.lo + b].sum()) < 1e-9
    # the rows outside [a, b) are exactly those that read a halo row
    nb = np.concatenate([d.klon.reshape(d.n, -1), d.klat.reshape(d.n, -1), d.kcor.reshape(d.n, -1)], 1)
    needs = ((nb >= d.n) & (nb < d.nland)).any(1)
    ok = ok and not needs[a:b].any()
t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(t) == 1 else 3)
'''


def test_eight_rank_decomposition_and_grid_file():
    """What a SCALE run at N = 8 executes before its first step, on the CPU: the grid tables through the file rank 0 writes for the other
    ranks (grid.save_grid / load_grid: identical tables), the eight bands (mpdecomp.F90:58-100: equal counts, contiguous), every halo row
    owned by exactly the peer that lists it in its send table in the same order (mpexchng.F90:141-206), and the send tables -- built from
    the bands within one latitude row only -- equal to the ones a scan of all bands gives."""
    import tempfile

    from ecwam_amd import decomp, grid as G

    g0 = G.build_grid(96)
    path = os.path.join(tempfile.mkdtemp(), "grid.npz")
    G.save_grid(g0, path)
    g = G.load_grid(path)
    for k in G._ARRAYS:
        assert np.array_equal(getattr(g0, k), getattr(g, k)), k
    assert (g0.name, g0.ngy, g0.nsea, g0.xdella, g0.amosop) == (g.name, g.ngy, g.nsea, g.xdella, g.amosop)
    nr = 8
    doms = [decomp.local_domain(g, r, nr) for r in range(nr)]
    bounds = decomp.split_points(g.nsea, nr)
    assert sum(d.n for d in doms) == g.nsea and all(d.lo == bounds[d.rank] and d.hi == bounds[d.rank + 1] for d in doms)
    for d in doms:
        assert sorted(d.recv) == sorted(p for p in range(nr) if p != d.rank and d.rank in doms[p].send)
        covered = 0
        for peer, (dst0, cnt) in d.recv.items():
            assert np.array_equal(d.halo_global[dst0 - d.n: dst0 - d.n + cnt], doms[peer].send[d.rank] + doms[peer].lo), (d.rank, peer)
            covered += cnt
        assert covered == d.nh
        # the brute-force send table: what EVERY other band's stencil reads of this band
        for p in range(nr):
            if p == d.rank:
                continue
            ph = decomp._halo_global(g, int(bounds[p]), int(bounds[p + 1]))
            mine = ph[(ph >= d.lo) & (ph < d.hi)] - d.lo
            assert (mine.size == 0 and p not in d.send) or np.array_equal(mine, d.send[p]), (d.rank, p)
        a, b = d.interior()
        assert 0 <= a <= b <= d.n and (b - a) > 0.8 * d.n      # most of a band is advected behind the exchange


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_exchange_gloo(tmp_path, world):
    """N>1 path on CPU: `world` processes, gloo backend, the same HaloExchange object the GPU path uses, through the
    start / finish pair Wamintgr.propag overlaps with the interior, for full rows and for compact fast-wave rows."""
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER)
    port = str(29537 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_oracle_refraction_restatement_invariants():
    """CPU checks of the oracle's IREFRA /= 0 restatement (gradi / propdot / ctuwdrv / ctuw / propags2 general branch):
    with IREFRA = 0 the general routines reproduce the plain ones (weights bit for bit, F3 up to the summation order);
    with zero currents IREFRA = 2 has no frequency shift and no downwind weights; the LLCFLCUROFF second call masks
    exactly the points that failed and leaves the others' weights untouched."""
    import test_gpu_refraction as R
    from oracle.oracle import Oracle

    c = R._case("dp", 2, n_oct=12, nang=12, nred=25)
    g, n = c["g"], c["n"]
    o = Oracle(c["cfg"], "dp")
    z = np.zeros(n + 1)
    w0 = o.ctu_weights(g, c["cg"], 600.0)
    dot0 = o.propdot(g, 0, c["dep"], z, z, c["wn"], c["cg"], c["om"])
    wg = o.ctu_weights_gen(g, 0, c["cg"], c["om"], z, z, dot0, 600.0)
    for k in ("SUMWN", "WLONN", "WLATN", "WCORN", "WKPMN"):
        assert np.array_equal(w0[k], wg[k]), k
    f3, f3g = o.propags2(g, c["f1"], w0), o.propags2_gen(g, c["f1"], wg)
    assert np.abs(f3 - f3g).max() < 8 * np.finfo(float).eps * np.abs(f3).max()
    # zero currents: only upwind weights, no frequency shift
    dz = o.propdot(g, 2, c["dep"], z, z, c["wn"], c["cg"], c["om"])
    wz = o.ctu_weights_gen(g, 2, c["cg"], c["om"], z, z, dz, 600.0)
    assert not wz["WMPMN"].any() and not dz["THDC"].any()
    assert np.array_equal(wz["SUMWN"], w0["SUMWN"])
    # with currents: every weight in [0,1], the frequency shift is active, mass leaves only through land / the poles
    d2 = o.propdot(g, 2, c["dep"], c["u"], c["v"], c["wn"], c["cg"], c["om"])
    w2 = o.ctu_weights_gen(g, 2, c["cg"], c["om"], c["u"], c["v"], d2, 600.0)
    assert w2["NFAIL"] == 0 and w2["WMPMN"].max() > 0
    for k in ("SUMWN", "WLONN", "WLATN", "WCORN", "WKPMN", "WMPMN"):
        assert w2[k].min() >= 0 and w2[k].max() <= 1
    assert np.abs(d2["THDC"]).max() > 0 and np.abs(d2["SDOT"]).max() > 0
    # current gradients are limited to CURRENT_GRADIENT_MAX * COSPH (gradi.F90:219-226): |THDC| <= 2 * that bound / cos(lat)
    assert np.abs(d2["THDC"]).max() <= 4e-5
    # second call: masked points == points that failed first; unmasked points keep their weights
    cs = R._case("dp", 3, n_oct=12, nang=12, nred=25, delpro=900, cur_amp=6.0, smooth_depth=False)
    os_ = Oracle(cs["cfg"], "dp")
    ds = os_.propdot(cs["g"], 3, cs["dep"], cs["u"], cs["v"], cs["wn"], cs["cg"], cs["om"])
    a = os_.ctu_weights_gen(cs["g"], 3, cs["cg"], cs["om"], cs["u"], cs["v"], ds, 900.0, llcflcuroff=False)
    b = os_.ctu_weights_gen(cs["g"], 3, cs["cg"], cs["om"], cs["u"], cs["v"], ds, 900.0, llcflcuroff=True)
    assert a["NFAIL"] > 0 and np.array_equal(b["CURMASK"] == 0, a["FAIL"] == 1)
    keep = a["FAIL"] == 0
    for k in ("SUMWN", "WKPMN", "WMPMN", "WLONN"):
        assert np.array_equal(a[k][keep], b[k][keep]), k
    assert not b["WMPMN"][~keep].any()


def test_speed_build_of_the_oracle_agrees_with_the_strict_build():
    """bench.py times `Oracle(fast=True)` (same sources, -O3 -march=x86-64-v3, contraction on) as the CPU baseline; it must
    compute the same thing as the bit-reproducible build every parity test uses (differences: FMA contraction and
    vectorised summation order only)."""
    from oracle.oracle import Oracle

    cfg = Config(nang=24, nfre=36, nfre_red=29)
    case = H.make_point_case(200, cfg, "dp", spectra="mixed", seed=3)
    a = H.oracle_implsch(case, Oracle(cfg, "dp"))
    b = H.oracle_implsch(case, Oracle(cfg, "dp", fast=True))
    st = H.compare_implsch(a, b, case["tables"])
    assert st["mij_flips"] == 0 and st["fl1_max_rel_peak_all"] < 1e-10 and st["swh_max_rel"] < 1e-12, st


def test_oracle_regression_vectors():
    """tests/golden/oracle_regression_dp.npz (tests/diag/make_oracle_regression.py): frozen outputs of this repository's own oracle
    for small seeded cases.  Regression only -- they say nothing about the reference.  Tolerance 1e-9 of each array's scale: libm
    may differ between the machine that wrote them and the one that checks them."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
    import make_oracle_regression as M

    want = np.load(os.path.join(ROOT, "tests", "golden", "oracle_regression_dp.npz"))
    got = M.cases()
    assert set(want.files) == set(got)
    for k in want.files:
        a, b = np.asarray(want[k], dtype=np.float64), np.asarray(got[k], dtype=np.float64)
        assert a.shape == b.shape, k
        if k.endswith("_MIJ"):
            assert np.array_equal(a, b), k
        else:
            assert np.max(np.abs(a - b)) <= 1e-9 * max(np.max(np.abs(a)), 1e-300), k


@pytest.mark.parametrize("prec", ["sp", "dp"])
@pytest.mark.parametrize("flags", [{}, dict(llgcbz0=True, llnormagam=True)], ids=["set_a", "set_b"])
def test_nproma_blocked_cpu_variant_agrees_with_the_oracle(prec, flags):
    """bench.py times IMPLSCH on the CPU through the NPROMA-blocked variant (oracle/ora_implsch_blk.inc: the three hot routines with the
    point index innermost, vector libm, flush-to-zero in the speed build).  It is a timing baseline, not the parity oracle -- but it must
    compute the same thing: MIJ and XLLWS identical, spectra / forcing / fluxes within vector-libm rounding of the point-by-point oracle,
    in the exact and in the speed build, with a ragged last block; configurations it does not restate are declined."""
    from oracle.oracle import Oracle
    cfg = Config(nang=24, nfre=36, nfre_red=29, **flags)
    n = 3 * 32 + 7
    case = H.make_point_case(n, cfg, prec, spectra="mixed", seed=17)
    pr = case["props"]
    args = (case["FL1"], pr["WAVNUM"], pr["CGROUP"], pr["CINV"], pr["XK2CG"], pr["STOKFAC"], case["ENV"], case["FF"], case["INTF"])
    for fast in (False, True):
        o = Oracle(cfg, prec, fast=fast)
        ref, blk = o.implsch(*args), o.implsch_blocked(*args)
        assert blk is not None
        st = H.compare_implsch(ref, blk, case["tables"])
        assert st["mij_flips"] == 0 and st["xllws_bins_diff"] == 0, st
        tol = (2e-5, 1e-4) if prec == "sp" else (1e-12, 1e-10)
        assert st["fl1_max_rel_peak_all"] < tol[0] and st["ff_max_rel_all"] < tol[0] and st["intf_max_rel_all"] < tol[1], st
    o = Oracle(Config(nang=24, nfre=36, nfre_red=29, iphys=0), prec)
    assert o.implsch_blocked(*args) is None


def test_packed_weight_cpu_advection_is_bit_identical():
    """ora_propags2_w8 (bench.py's CPU timing variant of PROPAGS2: the eight weights packed as contiguous streams, unit-stride loops
    over M) returns the bits of ora_propags2, in the exact and in the speed build of the oracle (continents mask: land neighbours)."""
    from ecwam_amd import synthetic as syn
    from oracle.oracle import Oracle
    cfg = Config(nang=12, nfre=36, nfre_red=30, idelt=900, idelpro=900)
    g = G.build_grid(16, mask="continents")
    t = Tables(cfg, np.float64)
    rng = np.random.default_rng(8)
    for prec, fast in (("dp", False), ("sp", False), ("sp", True)):
        o = Oracle(cfg, prec, fast=fast)
        dt = H.np_dtype(prec)
        n = g.nsea
        depth = rng.uniform(20.0, 3000.0, n)
        pr = syn.depth_props(depth, Tables(cfg, dt), dt)
        cg = np.zeros((n + 1, 36), dt)
        cg[:n] = pr["CGROUP"]
        cg[n] = syn.depth_props(np.array([998.999]), Tables(cfg, dt), dt)["CGROUP"][0]
        w = o.ctu_weights(g, cg, 900.0)
        f1 = np.zeros((n + 1, 12, 36), dt)
        f1[:n] = rng.uniform(0.0, 1.0, (n, 12, 36))
        ref = o.propags2(g, f1, w)
        got = o.propags2_w8(g, f1, o.pack_w8(n, w))
        assert np.array_equal(ref, got), (prec, fast)
