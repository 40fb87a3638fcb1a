"""GPU test of the drop-in boundary seen from the reference's host language: a Fortran program fills YOWDRVTYPE-shaped
host types (chunked NPROMA x NCHNK arrays), calls ECWAM_HIP_SETUP and WAMINTGR_HIP (iso_c_binding -> C ABI -> HIP kernels)
with the WAMODEL date sequence, and must reproduce the Python/ctypes host driving the same C ABI bit for bit."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import harness as H
from ecwam_amd.tables import Config

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class _LocalGrid:
    """The Grid-shaped view of one rank's band (decomp.LocalDomain): local numbering, halo rows after the owned ones, land last."""

    def __init__(self, g, d):
        self.nsea, self.ngy, self.xdella = d.n, g.ngy, g.xdella
        self.kxlt, self.klon, self.klat, self.kcor = d.kxlt, d.klon, d.klat, d.kcor
        self.wlat, self.wcor = g.wlat[d.lo:d.hi], g.wcor[d.lo:d.hi]
        self.cosph, self.sinph, self.zdello, self.cosphm1_ext = g.cosph, g.sinph, g.zdello, d.cosphm1_ext


def _write_case(path, m, cfg, grid, nproma, nstep, obs=None, nosource=False, dom=None, proenvhalo=False):
    """dom: the rank's decomp.LocalDomain for a multi-rank case (grid = its _LocalGrid): the header carries rank / ranks / halo size /
    interior range, the lists of ECWAM_HIP_SET_DECOMPOSITION (1-based, the reference's NTOPE / IJTOPE / NIJSTART) end the file."""
    from ecwam_amd import lib as L, synthetic as syn

    t = m.t
    dt = m.npdt
    n = grid.nsea
    nchnk = (n + nproma - 1) // nproma
    params = L.make_params(t)
    tp, keep = L.make_tables(t)
    peers = sorted(set(dom.send) | set(dom.recv)) if dom is not None else []
    ia, ib = dom.interior() if dom is not None else (0, n)
    with open(path, "wb") as f:
        hdr = np.array([cfg.nang, cfg.nfre, cfg.nfre_red, nproma, nchnk, n, grid.ngy, cfg.idelt, cfg.idelpro,
                        np.dtype(dt).itemsize, nstep, ctypes.sizeof(params), len(keep), int(m.ifrelfmax),
                        int(m.delpro_lf or 0), int(cfg.irefra != 0), int(bool(getattr(m, 'llcflcuroff', False))),
                        int(obs is not None), int(nosource),
                        dom.nranks if dom is not None else 1, dom.rank if dom is not None else 0, dom.nh if dom is not None else 0,
                        len(peers), ia + 1, ib, int(proenvhalo)], dtype=np.int32)
        f.write(hdr.tobytes())
        f.write(bytes(params))
        for a in keep:
            f.write(np.int32(a.size).tobytes())
            f.write(np.ascontiguousarray(a).tobytes())

        def F(a, dtype):  # Fortran-order bytes
            return np.asfortranarray(np.asarray(a, dtype=dtype)).tobytes(order="F")

        f.write(F(grid.klon + 1, np.int32) + F(grid.klat + 1, np.int32) + F(grid.kcor + 1, np.int32) + F(grid.kxlt + 1, np.int32))
        f.write(F(grid.wlat, dt) + F(grid.wcor, dt) + F(grid.zdello, dt) + dt(grid.xdella).tobytes() + F(grid.cosph, dt) + F(grid.sinph, dt))
        f.write(F(m.cgroup_ext.cpu().numpy(), dt) + F(grid.cosphm1_ext, dt))

        def chunk(a):  # [n][...] point-major -> (NPROMA, ..., NCHNK) with pad lanes replicating lane 1
            pad = nchnk * nproma - n
            a = np.concatenate([a, np.repeat(a[(nchnk - 1) * nproma:(nchnk - 1) * nproma + 1], pad, axis=0)], 0) if pad else a
            a = a.reshape((nchnk, nproma) + a.shape[1:])
            return np.moveaxis(a, (0, 1), (-1, 0))  # (NPROMA, ..., NCHNK)

        fl = m.fl1.cpu().numpy()[:n]
        f.write(F(chunk(fl), dt))
        wv = m.wvprpt.cpu().numpy()
        for i in range(5):
            f.write(F(chunk(wv[:, i]), dt))
        ff = m.ff.cpu().numpy()
        for i in range(16):
            f.write(F(chunk(ff[:, i]), dt))
        if cfg.irefra and not proenvhalo:
            for a in (m.depth_ext, m.u_ext, m.v_ext, m.omosnh2kd_ext, m.wavnum_ext):
                f.write(F(a.cpu().numpy(), dt))
        elif cfg.irefra:       # this rank's own OMOSNH2KD / UCUR / VCUR (chunked) and the land row: ECWAM_HIP_PROENVHALO assembles the rest
            f.write(F(chunk(m.omosnh2kd_ext.cpu().numpy()[:n]), dt) + F(chunk(m.u_ext.cpu().numpy()[:n]), dt) + F(chunk(m.v_ext.cpu().numpy()[:n]), dt))
            land = syn.depth_props(np.array([float(t.BATHYMAX)]), t, dt)
            f.write(np.concatenate([land["WAVNUM"][0], land["CGROUP"][0], land["OMOSNH2KD"][0], [float(t.BATHYMAX), 0.0, 0.0]]).astype(dt).tobytes())
        if obs is not None:   # OBS[n][8][NFRE] -> OBSLAT(N,NFRE_RED,2), OBSLON(N,NFRE_RED,2), OBSCOR(N,NFRE_RED,4)
            o = np.moveaxis(np.asarray(obs)[:, :, :cfg.nfre_red], 1, 2)
            f.write(F(o[:, :, 0:2], dt) + F(o[:, :, 2:4], dt) + F(o[:, :, 4:8], dt))
        if dom is not None:
            sc = np.array([len(dom.send.get(p_, ())) for p_ in peers], np.int32)
            f.write(np.array(peers, np.int32).tobytes() + sc.tobytes())
            f.write(np.array([dom.recv.get(p_, (0, 0))[0] + 1 for p_ in peers], np.int32).tobytes())
            f.write(np.array([dom.recv.get(p_, (0, 0))[1] for p_ in peers], np.int32).tobytes())
            f.write(np.concatenate([np.asarray(dom.send[p_], np.int32) + 1 for p_ in peers if p_ in dom.send]).tobytes())
    return nchnk


def _read_out(path, dt, nproma, nchnk, nang, nfre, n, nemo=None):
    """The harness' output file -> point-major arrays of the n owned points."""
    raw = np.fromfile(path, dtype=np.uint8)
    nsp = nproma * nang * nfre * nchnk
    isz = np.dtype(dt).itemsize
    off = 0
    fl_f = np.frombuffer(raw, dtype=dt, count=nsp, offset=off).reshape((nproma, nang, nfre, nchnk), order="F"); off += nsp * isz
    xl_f = np.frombuffer(raw, dtype=dt, count=nsp, offset=off).reshape((nproma, nang, nfre, nchnk), order="F"); off += nsp * isz
    mij_f = np.frombuffer(raw, dtype=np.int32, count=nproma * nchnk, offset=off).reshape((nproma, nchnk), order="F"); off += 4 * nproma * nchnk
    ff_f = np.frombuffer(raw, dtype=dt, count=14 * nproma * nchnk, offset=off).reshape((nproma, nchnk, 14), order="F"); off += 14 * nproma * nchnk * isz
    in_f = np.frombuffer(raw, dtype=dt, count=15 * nproma * nchnk, offset=off).reshape((nproma, nchnk, 15), order="F"); off += 15 * nproma * nchnk * isz
    ij = np.arange(n)
    out = dict(FL1=fl_f[ij % nproma, :, :, ij // nproma], XLLWS=xl_f[ij % nproma, :, :, ij // nproma], MIJ=mij_f[ij % nproma, ij // nproma],
               FF=ff_f[ij % nproma, ij // nproma, :], INTF=in_f[ij % nproma, ij // nproma, :], raw_fl=fl_f)
    if (off < raw.size) if nemo is None else nemo:      # LWNEMOCOU: NEMONTAU and the 13 WAVE2OCEAN members (double), in the order of the device rows
        out["NEMONTAU"] = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off)[0]); off += 4
        w = np.frombuffer(raw, dtype=np.float64, count=13 * nproma * nchnk, offset=off).reshape((nproma, nchnk, 13), order="F")
        out["W2N"] = w[ij % nproma, ij // nproma, :]
        off += 8 * 13 * nproma * nchnk
    out["_rest"] = raw[off:]
    return out


@pytest.mark.parametrize("prec,lf,irefra,subgrid,nosource", [("sp", 0, 0, False, False), ("dp", 0, 0, False, False), ("sp", 5, 0, False, False),
                                                             ("sp", 0, 2, False, False), ("dp", 0, 3, False, False), ("sp", 0, 0, True, False),
                                                             ("sp", 0, 2, True, False), ("sp", 0, 0, False, True), ("dp", 5, 0, False, True),
                                                             ("sp", 5, 2, False, False), ("dp", 4, 3, False, False)])
def test_fortran_wamintgr_hip_matches_python_host(tmp_path, prec, lf, irefra, subgrid, nosource):
    _fortran_vs_python_host(tmp_path, prec, lf, irefra, subgrid, nosource)


@pytest.mark.parametrize("prec,nosource,lf,subgrid", [("sp", False, 0, False), ("dp", False, 0, False), ("sp", True, 0, False), ("sp", False, 5, False),
                                                      ("dp", False, 4, False), ("sp", False, 0, True), ("dp", False, 5, True)])
def test_fortran_one_kernel_step_matches_python_host_two_kernels(tmp_path, prec, nosource, lf, subgrid):
    """36 directions: WAMINTGR_HIP takes the one-kernel step (ecwam_hip_propags2_implsch: the exchange posted at propagation time, PROPAGS2
    inside IMPLSCH's tile load when the source terms are due), the Python host drives PROPAGS2 and IMPLSCH as two kernels -- the same bits.
    With LLSOURCE = F the Fortran side must fall back to the separate advection.  lf > 0: fast-wave sub-steps on compact rows, the last one
    inside the kernel."""
    _fortran_vs_python_host(tmp_path, prec, lf, 0, subgrid, nosource, nang=36, nfre_red=36, idelt=450)      # subgrid: LSUBGRID (ECWAM_HIP_SET_SUBGRID)


def _fortran_vs_python_host(tmp_path, prec, lf, irefra, subgrid, nosource, nang=12, nfre_red=25, idelt=900):
    """lf > 0 together with irefra > 0: fast-wave sub-steps with refraction (propag_wam.F90:175-212 + :247-313), the reference's call
    sequence on the full rows with one CURMASK per frequency range.
    nosource: YOWSTAT's LLSOURCE = F -- the branch of wamintgr.F90:152-160 (FL1 = MAX(FL1, EPSMIN), MIJ = NFRE, XLLWS = 0) runs on the
    device copies (ecwam_hip_nosource); the advected spectra still hold exact zeros from the land neighbours before the clamp."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import build, grid as G
    from ecwam_amd.wamintgr import Wamintgr

    exe = build.fortran_exe(prec)
    if not os.path.exists(exe):
        build.build_fortran()
    cfg = Config(nang=nang, nfre=36, nfre_red=nfre_red, idelt=idelt, idelpro=idelt, irefra=irefra)
    g = G.build_grid(16, mask="continents")
    m = Wamintgr(cfg, g, prec, ifrelfmax=lf, delpro_lf=idelt / 2.0 if lf else None)    # lf: fast waves M <= lf in two sub-steps
    m.init_synthetic(seed=21)
    m.llcflcuroff = irefra != 3     # refraction: ECWAM_HIP_SET_ENVIRONMENT on the Fortran side, with and without LLCFLCUROFF
    assert m.nrows == g.nsea + 1
    obs = None
    if subgrid:     # LSUBGRID: ECWAM_HIP_SET_SUBGRID on the Fortran side
        from ecwam_amd import synthetic as syn
        obs = syn.obstructions(g, cfg.nfre, seed=5)
        obs[:, :, cfg.nfre_red:] = 1.0
        m.set_obstructions(obs)
    nproma, nstep = 24, 2
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    nchnk = _write_case(case, m, cfg, g, nproma, nstep, obs, nosource=nosource)
    r = subprocess.run([exe, case, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    for _ in range(nstep):
        m.step(llsource=not nosource)
    torch.cuda.synchronize()
    if nosource:
        epsmin = float(m.t.EPSMIN)
        assert float(m.fl1[: g.nsea].min()) >= epsmin and int(m.mij.min()) == cfg.nfre == int(m.mij.max()) and float(m.xllws.abs().max()) == 0.0
    dt = m.npdt
    n = g.nsea
    o = _read_out(out, dt, nproma, nchnk, cfg.nang, cfg.nfre, n)
    fl_f = o["raw_fl"]
    assert np.array_equal(o["FL1"], m.fl1.cpu().numpy()[:n])
    assert np.array_equal(o["XLLWS"], m.xllws.cpu().numpy())
    assert np.array_equal(o["MIJ"], m.mij.cpu().numpy())
    assert np.array_equal(o["FF"], m.ff.cpu().numpy()[:, :14])
    # INTGT_PARAM_FIELDS members written in the order of synthetic.INTF_NAMES
    assert np.array_equal(o["INTF"][:, [2, 3, 5, 6, 7, 8, 9, 12, 13, 14]], m.intf.cpu().numpy()[:, [2, 3, 5, 6, 7, 8, 9, 12, 13, 14]])
    # pad lanes of the ragged last chunk replicate its lane 1 (propag_wam.F90:388-398)
    kl = n - (nchnk - 1) * nproma
    if kl < nproma:
        assert np.array_equal(fl_f[kl:, :, :, -1], np.repeat(fl_f[:1, :, :, -1], nproma - kl, axis=0))
    m.ctx.close()


@pytest.mark.parametrize("prec,lf,irefra,nang", [("sp", 0, 0, 12), ("dp", 5, 0, 12), ("sp", 0, 2, 12), ("dp", 4, 3, 12), ("sp", 0, 0, 36)])
def test_fortran_two_processes_on_one_gpu_match_single_domain(tmp_path, prec, lf, irefra, nang):
    """The multi-rank path of the Fortran layer, executed: two processes of the harness share the GPU, each owns one band of the sea
    points (ECWAM_HIP_SET_DECOMPOSITION with the reference's 1-based NTOPE / IJTOPE / NIJSTART-style lists -> 0-based device rows,
    halo rows and land slot behind the owned rows, interior range), the halo travels host-staged (ecwam_hip_halo_pack_host ->
    the harness' exchange through files, where ecWAM would call MPI -> ecwam_hip_halo_unpack_host), fast-wave sub-steps exchange the
    compact rows.  Both bands together must equal the single-domain Python host bit for bit.  Also LWNEMOCOU: the WAVE2OCEAN sums and
    the accumulation count NEMONTAU (wamintgr.F90:150) after two source steps.  irefra > 0: refraction on two ranks -- each process hands
    ECWAM_HIP_PROENVHALO its own DEPTH / UCUR / VCUR / OMOSNH2KD, the halo rows of the environment travel through the same exchange as rows
    of 3 NFRE + 3 reals, every PROPAGS2 call of the (sub-stepped) sequence runs behind its own exchange of the spectra.
    nang = 36: the Fortran ranks take the one-kernel step (interior rows behind the posted exchange, the two ends behind its arrival)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import build, decomp, grid as G
    from ecwam_amd.wamintgr import Wamintgr

    exe = build.fortran_exe(prec)
    if not os.path.exists(exe):
        build.build_fortran()
    dt_ = 900 if nang == 12 else 450
    cfg = Config(nang=nang, nfre=36, nfre_red=25 if nang == 12 else 36, idelt=dt_, idelpro=dt_, lwnemocou=True, irefra=irefra)
    g = G.build_grid(16, mask="continents")
    kw = dict(ifrelfmax=lf, delpro_lf=450.0 if lf else None)
    nranks, nproma, nstep = 2, 24, 2
    xdir = tmp_path / "xchg"
    xdir.mkdir()
    procs, meta = [], []
    for r in range(nranks):
        mr = Wamintgr(cfg, g, prec, rank=r, nranks=nranks, halo_transport="host", **kw)
        mr.init_synthetic(seed=21)
        case, out = str(tmp_path / f"case{r}.bin"), str(tmp_path / f"out{r}.bin")
        mr.llcflcuroff = irefra != 3
        nchnk = _write_case(case, mr, cfg, _LocalGrid(g, mr.dom), nproma, nstep, dom=mr.dom, proenvhalo=bool(irefra))
        meta.append((out, nchnk, mr.dom.n))
        mr.ctx.close()
        procs.append(subprocess.Popen([exe, case, out, "run", str(xdir)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and "ok" in o, o
    m = Wamintgr(cfg, g, prec, **kw)
    m.init_synthetic(seed=21)
    m.llcflcuroff = irefra != 3
    w2n = torch.zeros((g.nsea, 13), dtype=torch.float64, device=m.dev)
    for _ in range(nstep):
        m.propag()
        m.newwind()
        m.implsch(wam2nemo=w2n)
    torch.cuda.synchronize()
    outs = [_read_out(o, m.npdt, nproma, nc, cfg.nang, cfg.nfre, n) for o, nc, n in meta]
    n = g.nsea
    cat = lambda k: np.concatenate([o[k] for o in outs])
    assert np.array_equal(cat("FL1"), m.fl1.cpu().numpy()[:n])
    assert np.array_equal(cat("XLLWS"), m.xllws.cpu().numpy())
    assert np.array_equal(cat("MIJ"), m.mij.cpu().numpy())
    assert np.array_equal(cat("FF"), m.ff.cpu().numpy()[:, :14])
    assert all(o["NEMONTAU"] == nstep for o in outs)
    got, want = cat("W2N"), w2n.cpu().numpy()
    assert np.array_equal(got, want) and np.abs(want[:, 7]).max() > 0        # NEMOTAUX accumulated over the two steps
    m.ctx.close()


@pytest.mark.parametrize("prec,lf", [("sp", 0), ("dp", 5)])
def test_fortran_rccl_transport_with_a_self_peer_matches_single_domain(tmp_path, prec, lf):
    """The RCCL path of the Fortran layer executed on the one GPU there is (mpexchng.F90:164-206 as the library's grouped ncclSend / ncclRecv):
    ECWAM_HIP_COMM_UNIQUE_ID -> ECWAM_HIP_SET_DECOMPOSITION with the 128-byte id and NO host exchange routine -> ECWAM_HIP_SETUP
    (ecwam_hip_halo_setup, ecwam_hip_comm_init) -> WAMINTGR_HIP steps whose HIP_HALO_START / HIP_HALO_FINISH post and await the exchange
    around the interior advection.  One rank that is its own peer: the last rows of the grid are mirrored into halo rows and every
    neighbour reference to them is redirected to the mirror, so each PROPAGS2 (and each fast-wave sub-step on the compact rows) reads
    those spectra only through the exchange -- and the result must equal the plain single-domain run bit for bit.  In a child process
    under a timeout: a transport that hangs fails the test, not the session."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import types

    from ecwam_amd import build, grid as G
    from ecwam_amd.wamintgr import Wamintgr

    exe = build.fortran_exe(prec)
    if not os.path.exists(exe):
        build.build_fortran()
    cfg = Config(nang=12, nfre=36, nfre_red=25, idelt=900, idelpro=900)
    g = G.build_grid(16, mask="continents")
    kw = dict(ifrelfmax=lf, delpro_lf=450.0 if lf else None)
    nproma, nstep = 24, 2
    m = Wamintgr(cfg, g, prec, **kw)
    m.init_synthetic(seed=21)
    n = g.nsea
    nh = 57                                         # the mirrored rows: the last 57 sea points (more than one latitude row)
    src = np.arange(n - nh, n)

    def redirect(a):                                 # single-domain numbering (land = n) -> owned [0, n), mirrors [n, n + nh), land n + nh
        a = np.asarray(a).astype(np.int64).copy()
        land, mirrored = a == g.nland, (a >= n - nh) & (a < n)
        a[mirrored] += nh
        a[land] = n + nh
        return a.astype(np.int32)

    klon, klat, kcor = redirect(g.klon), redirect(g.klat), redirect(g.kcor)
    reads_halo = ((klon >= n) & (klon < n + nh)).reshape(n, -1).any(1) | ((klat >= n) & (klat < n + nh)).reshape(n, -1).any(1) | \
                 ((kcor >= n) & (kcor < n + nh)).reshape(n, -1).any(1)
    ib = int(np.flatnonzero(reads_halo).min())
    assert 0 < ib < n and reads_halo.sum() >= nh
    cm1 = np.asarray(g.cosphm1_ext)
    dom = types.SimpleNamespace(n=n, nh=nh, rank=0, nranks=1, lo=0, hi=n, send={0: src.astype(np.int32)}, recv={0: (n, nh)}, kxlt=g.kxlt,
                                klon=klon, klat=klat, kcor=kcor, cosphm1_ext=np.concatenate([cm1[:n], cm1[src], cm1[n:n + 1]]),
                                interior=lambda: (0, ib))
    cg = m.cgroup_ext.cpu().numpy()
    shim = types.SimpleNamespace(t=m.t, npdt=m.npdt, fl1=m.fl1, wvprpt=m.wvprpt, ff=m.ff, ifrelfmax=m.ifrelfmax, delpro_lf=m.delpro_lf,
                                 cgroup_ext=torch.from_numpy(np.concatenate([cg[:n], cg[src], cg[n:n + 1]])))
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    nchnk = _write_case(case, shim, cfg, _LocalGrid(g, dom), nproma, nstep, dom=dom)
    r = subprocess.run([exe, case, out, "run", "rccl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout
    for _ in range(nstep):
        m.step()
    torch.cuda.synchronize()
    got = _read_out(out, m.npdt, nproma, nchnk, cfg.nang, cfg.nfre, n)
    assert np.array_equal(got["FL1"], m.fl1.cpu().numpy()[:n])
    assert np.array_equal(got["MIJ"], m.mij.cpu().numpy()) and np.array_equal(got["XLLWS"], m.xllws.cpu().numpy())
    assert np.array_equal(got["FF"], m.ff.cpu().numpy()[:, :14])
    m.ctx.close()


@pytest.mark.parametrize("prec,nemo", [("sp", True), ("dp", True), ("sp", False)])
def test_fortran_seam_sequence_replays_the_reference_call_lines(tmp_path, prec, nemo):
    """The complete FIELD_API call sequence of the reference's GPU build around the seam (ecwam_amd/fortran/seam_sequence.F90: the lines of
    wamodel.F90:207-226,376-385,435-470,614-642,651-671 and wamintgr_loki_gpu.F90:100-157,197-200 as they stand, WAMINTGR_HIP in the place
    of WAMINTGR_LOKI_GPU) on the host types of yowdrvtype_hip.F90: initial copies on queues 1-3, per-step copies back on queues 4 (FL1,
    FF_NOW), 5 (WVENVI) and 6 (WAM2NEMO), an output step, a restart step, two NEMO coupling steps (UPDNEMOSTRESS averages and resets the
    accumulated stresses on the host, SYNC_DEVICE_RDWR sends them back on queue 3), new winds through FF_NEXT, the final
    GET_HOST_DATA_RDWR / DELETE_DEVICE_DATA.  Every host array it ends with, the spectra it read at the output step and the stresses it
    handed to 'NEMO' equal the Python host driving the same C ABI bit for bit; the harness itself stops if a call copies something the
    other side did not change (device -> host only for what the device wrote)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ecwam_amd import build, grid as G
    from ecwam_amd.wamintgr import Wamintgr

    exe = build.fortran_exe(prec, "seam_sequence")
    if not os.path.exists(exe):
        build.build_fortran()
    cfg = Config(nang=12, nfre=36, nfre_red=25, idelt=900, idelpro=900, lwnemocou=nemo)
    g = G.build_grid(16, mask="continents")
    m = Wamintgr(cfg, g, prec)
    m.init_synthetic(seed=33)
    nproma, nstep, n = 24, 4, g.nsea
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    nchnk = _write_case(case, m, cfg, g, nproma, nstep)
    rst = str(tmp_path / "BLS_restart")
    r = subprocess.run([exe, case, out, rst], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "seam_sequence: ok" in r.stdout, r.stdout + r.stderr
    ffn0 = m.ff.clone()
    w2n = torch.zeros((n, 13), dtype=torch.float64, device=m.dev) if nemo else None
    acc = [7, 8, 11, 12, 9, 10]      # NEMOTAUX, NEMOTAUY, NEMOWSWAVE, NEMOPHIF, NEMOTAUICX, NEMOTAUICY in the device rows
    avgs, fl_out, ufric_out = [], None, None
    for k in range(1, nstep + 1):
        m.ff_next = None
        if k == 3:      # new winds in FF_NEXT, handed over by NEWWIND inside this step
            ffn = ffn0.clone()
            ffn[:, 3] = ffn[:, 3] * 1.1
            ffn[:, 1] = ffn[:, 1] + 0.3
            ffn[:, 2] = ffn[:, 2] * 0.5
            m.ff_next = ffn
        m.propag()
        m.newwind()
        m.implsch(wam2nemo=w2n)
        if k == 2:
            fl_out, ufric_out = m.fl1[:n].cpu().numpy().copy(), m.ff[:, 7].cpu().numpy().copy()
        if nemo and k % 2 == 0:      # UPDNEMOSTRESS: average over the two steps, reset
            avgs.append((w2n[:, acc] * 0.5).cpu().numpy())
            w2n[:, acc] = 0.0
    torch.cuda.synchronize()
    dt = m.npdt
    o = _read_out(out, dt, nproma, nchnk, cfg.nang, cfg.nfre, n, nemo=nemo)
    assert np.array_equal(o["FL1"], m.fl1.cpu().numpy()[:n])
    assert np.array_equal(o["XLLWS"], m.xllws.cpu().numpy())
    assert np.array_equal(o["MIJ"], m.mij.cpu().numpy())
    assert np.array_equal(o["FF"], m.ff.cpu().numpy()[:, :14])
    assert np.array_equal(o["INTF"], m.intf.cpu().numpy()[:, :15])
    rest = o["_rest"]
    ij = np.arange(n)
    off = 0
    if nemo:
        assert o["NEMONTAU"] == nstep and np.array_equal(o["W2N"], w2n.cpu().numpy())
    ncoup = int(np.frombuffer(rest, dtype=np.int32, count=1, offset=off)[0]); off += 4
    assert ncoup == (2 if nemo else 0)
    if ncoup:
        sa = np.frombuffer(rest, dtype=np.float64, count=nproma * nchnk * 6 * ncoup, offset=off).reshape((nproma, nchnk, 6, ncoup), order="F")
        off += 8 * nproma * nchnk * 6 * ncoup
        for c in range(ncoup):
            assert np.array_equal(sa[ij % nproma, ij // nproma, :, c], avgs[c]) and np.abs(avgs[c][:, 0]).max() > 0
    nsp = nproma * cfg.nang * cfg.nfre * nchnk
    isz = np.dtype(dt).itemsize
    fl_f = np.frombuffer(rest, dtype=dt, count=nsp, offset=off).reshape((nproma, cfg.nang, cfg.nfre, nchnk), order="F"); off += nsp * isz
    uf_f = np.frombuffer(rest, dtype=dt, count=nproma * nchnk, offset=off).reshape((nproma, nchnk), order="F"); off += nproma * nchnk * isz
    assert np.array_equal(fl_f[ij % nproma, :, :, ij // nproma], fl_out)        # what the host read at its output step (queue 4)
    assert np.array_equal(uf_f[ij % nproma, ij // nproma], ufric_out)
    # the restart file ECWAM_HIP_WRITEFL wrote at the last step (writefl.F90:110-118: one unformatted record (((FL(IJ,K,M),IJ),K),M);
    # a second one in the re-labelled order of a 2-D decomposition, here the points reversed), read with the Python host's reader
    from ecwam_amd import restart
    want = m.fl1.cpu().numpy()[:n]
    assert np.array_equal(restart.read_fl(rst, n, cfg.nang, cfg.nfre, dt, record=0), want)
    assert np.array_equal(restart.read_fl(rst, n, cfg.nang, cfg.nfre, dt, record=1), want[::-1])
    nh2d, nd2h, bh2d, bd2h = (int(x) for x in np.frombuffer(rest, dtype=np.int64, count=4, offset=off))
    # FL1 travels up once (+ the five FREQUENCY members: 5/12 of its size here, + the per-point fields) and comes down once per step;
    # XLLWS once (the test's own request at the end)
    assert nsp * isz <= bh2d < 2 * nsp * isz and (nstep + 1) * nsp * isz <= bd2h < (nstep + 2) * nsp * isz, (nh2d, nd2h, bh2d, bd2h)
    m.ctx.close()
