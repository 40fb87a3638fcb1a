"""ctypes wrapper of the CPU oracle (oracle/*.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package ecwam_amd never does).  "Parity unpinned": see oracle/ora.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OraCfg(C.Structure):
    _fields_ = [
        ("nang", C.c_int), ("nfre", C.c_int), ("nfre_red", C.c_int), ("ifre1", C.c_int), ("fr1", C.c_double),
        ("idelt", C.c_int), ("idelpro", C.c_int), ("ximp", C.c_double),
        ("iphys", C.c_int), ("isnonlin", C.c_int), ("irefra", C.c_int), ("icode", C.c_int),
        ("llgcbz0", C.c_int), ("llnormagam", C.c_int), ("llcapchnk", C.c_int),
        ("lbiwbk", C.c_int), ("licerun", C.c_int), ("lmaskice", C.c_int), ("lwamrsetci", C.c_int),
        ("lciwa1", C.c_int), ("lciwa2", C.c_int), ("lciwa3", C.c_int), ("lciscal", C.c_int),
        ("lwvflx_snl", C.c_int), ("lwflux", C.c_int), ("lwfluxout", C.c_int), ("lwnemocou", C.c_int),
        ("lwcou", C.c_int), ("lwcouast", C.c_int),
        ("lwnemocouwrs", C.c_int), ("lwnemocouibr", C.c_int), ("lwnemotauoc", C.c_int), ("lwnemocousend", C.c_int),
        ("lwnemocoustk", C.c_int),
        ("wspmin", C.c_double), ("rnu", C.c_double), ("rnum", C.c_double),
        ("lwnemocoustrn", C.c_int), ("zalpfacb", C.c_double), ("zalpfacx", C.c_double), ("zalpwrs", C.c_double),
        ("zibrw_thrsh", C.c_double),
    ]


def build(force: bool = False) -> None:
    libs = ("libora_sp.so", "libora_dp.so", "libora_sp_fast.so", "libora_dp_fast.so")
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in libs)
    if not need:
        src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile")
        so_m = min(os.path.getmtime(os.path.join(_HERE, f)) for f in libs)
        need = src_m > so_m
    if need:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)


class Oracle:
    """One initialised oracle instance (module state lives inside the shared object, like the
    Fortran modules it restates: one configuration per loaded library)."""

    def __init__(self, cfg, precision: str = "dp", fast: bool = False):
        """fast=True loads the speed build (vectorised, contraction on): for the CPU timing leg of bench.py only; every
        parity check uses the default bit-reproducible build."""
        build()
        self.precision = precision
        self.dtype = np.float32 if precision == "sp" else np.float64
        # a private copy per instance so that sp/dp and several configs can coexist in one process
        path = os.path.join(_HERE, f"libora_{precision}{'_fast' if fast else ''}.so")
        self._fresh_copy(path)
        assert self.lib.ora_real_size() == np.dtype(self.dtype).itemsize
        oc = OraCfg()
        self.lib.ora_default_cfg(C.byref(oc))
        for name, _ in OraCfg._fields_:
            if hasattr(cfg, name):
                setattr(oc, name, type(getattr(oc, name))(getattr(cfg, name)))
        rc = self.lib.ora_init(C.byref(oc))
        if rc:
            raise RuntimeError(f"ora_init failed rc={rc}")
        if getattr(cfg, "lciwa1", False):   # SDICE1's tabulated block (cigetdeac.F90:85-552): the oracle's own copy of the data table
            raw = np.ascontiguousarray(np.loadtxt(os.path.join(_HERE, "data", "cideac_kohout_meylan.txt")), dtype=np.float64)
            assert raw.shape == (36, 11)
            self.lib.ora_set_cideac(raw.ctypes.data_as(C.c_void_p))
        self.cfg = cfg
        self.NANG, self.NFRE = cfg.nang, cfg.nfre
        self.NFRE_RED = cfg.nfre_red if cfg.nfre_red > 0 else cfg.nfre
        self.lib.ora_get.restype = C.c_int

    def _fresh_copy(self, path: str) -> None:
        import shutil
        import tempfile

        tmp = tempfile.NamedTemporaryFile(prefix="libora_", suffix=".so", delete=False)
        tmp.close()
        shutil.copy(path, tmp.name)
        self.lib = C.CDLL(tmp.name, mode=os.RTLD_LOCAL)
        os.unlink(tmp.name)

    # -- helpers
    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def get(self, name: str) -> np.ndarray:
        buf = np.zeros(8192, dtype=np.float64)
        n = self.lib.ora_get(name.encode(), self._p(buf), C.c_int(buf.size))
        if n < 0:
            raise KeyError(name)
        return buf[:n].copy()

    def depthprpt(self, depth: np.ndarray) -> dict:
        n = depth.size
        d = np.ascontiguousarray(depth, dtype=self.dtype)
        out = {k: np.zeros((n, self.NFRE), dtype=self.dtype) for k in ("WAVNUM", "CINV", "CGROUP", "XK2CG", "OMOSNH2KD", "STOKFAC")}
        out["EMAXDPT"] = np.zeros(n, dtype=self.dtype)
        self.lib.ora_depthprpt(C.c_int(n), self._p(d), self._p(out["WAVNUM"]), self._p(out["CINV"]), self._p(out["CGROUP"]),
                               self._p(out["XK2CG"]), self._p(out["OMOSNH2KD"]), self._p(out["STOKFAC"]), self._p(out["EMAXDPT"]))
        return out

    def implsch(self, fl1, wavnum, cgroup, cinv, xk2cg, stokfac, env, ff, intf, want_dbg=False, w2n=None, ibrmem=None):
        """All arrays are copied; returns dict(FL1, XLLWS, MIJ, FF, INTF[, DBG]).  ibrmem: ENVIRONMENT%IBRMEM [n] (LWNEMOCOUIBR)."""
        n = fl1.shape[0]
        T = self.dtype
        ib = None if ibrmem is None else np.ascontiguousarray(ibrmem, dtype=T)
        self.lib.ora_set_ibrmem(None if ib is None else self._p(ib))
        fl1 = np.array(fl1, dtype=T, order="C")
        ff = np.array(ff, dtype=T, order="C")
        intf = np.array(intf, dtype=T, order="C")
        xllws = np.zeros_like(fl1)
        mij = np.zeros(n, dtype=np.int32)
        dbg = np.zeros((n, 8), dtype=T) if want_dbg else None
        a = [np.ascontiguousarray(x, dtype=T) for x in (wavnum, cgroup, cinv, xk2cg, stokfac, env)]
        if w2n is not None:
            w2n = np.array(w2n, dtype=np.float64, order="C")
            assert w2n.shape == (n, 13)
        rc = self.lib.ora_implsch_w2n(C.c_int(n), self._p(fl1), *(self._p(x) for x in a[:5]), self._p(a[5]), self._p(ff),
                                      self._p(intf), self._p(mij), self._p(xllws), self._p(dbg) if want_dbg else None,
                                      None if w2n is None else w2n.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(f"ora_implsch abort branch rc={rc}")
        out = dict(FL1=fl1, XLLWS=xllws, MIJ=mij, FF=ff, INTF=intf)
        if w2n is not None:
            out["W2N"] = w2n
        if want_dbg:
            out["DBG"] = dbg
        return out

    def implsch_blocked(self, fl1, wavnum, cgroup, cinv, xk2cg, stokfac, env, ff, intf):
        """The NPROMA-blocked variant (ora_implsch_blk.inc: SINPUT_ARD / SDISSIP_ARD / SNONLIN with the point index innermost and
        vector libm): the CPU timing baseline.  Same arguments and result as implsch(); None when the configuration is outside
        what the blocked routines restate."""
        n = fl1.shape[0]
        T = self.dtype
        fl1 = np.array(fl1, dtype=T, order="C")
        ff = np.array(ff, dtype=T, order="C")
        intf = np.array(intf, dtype=T, order="C")
        xllws = np.zeros_like(fl1)
        mij = np.zeros(n, dtype=np.int32)
        a = [np.ascontiguousarray(x, dtype=T) for x in (wavnum, cgroup, cinv, xk2cg, stokfac, env)]
        rc = self.lib.ora_implsch_blk(C.c_int(n), self._p(fl1), *(self._p(x) for x in a[:5]), self._p(a[5]), self._p(ff), self._p(intf),
                                      self._p(mij), self._p(xllws))
        if rc == -1:
            return None
        if rc:
            raise RuntimeError(f"ora_implsch_blk abort branch rc={rc}")
        return dict(FL1=fl1, XLLWS=xllws, MIJ=mij, FF=ff, INTF=intf)

    def outbs(self, fl1, zmiss=-999.0):
        """OUTBLOCK parameters 1-3 and 6: returns [n][5] = SWH, MWD [deg], MWP (or zmiss), EM, PP1D (or zmiss)."""
        fl1 = np.ascontiguousarray(fl1, dtype=self.dtype)
        n = fl1.shape[0]
        out = np.zeros((n, 5), dtype=self.dtype)
        zm = C.c_float(zmiss) if self.dtype == np.float32 else C.c_double(zmiss)
        self.lib.ora_outbs(C.c_int(n), self._p(fl1), zm, self._p(out))
        return out

    def newwind(self, ff, ffn):
        ff = np.array(ff, dtype=self.dtype, order="C")
        ffn = np.ascontiguousarray(ffn, dtype=self.dtype)
        self.lib.ora_newwind(C.c_int(ff.shape[0]), self._p(ff), self._p(ffn))
        return ff

    def set_obstructions(self, obs):
        """LSUBGRID: OBS[n][8][NFRE] (OBSLAT 1:2, OBSLON 1:2, OBSCOR 1:4 per frequency) for the following ctu_weights* calls."""
        self._obs = None if obs is None else np.ascontiguousarray(obs, dtype=self.dtype)
        self.lib.ora_set_obstructions(None if self._obs is None else self._p(self._obs))

    def ctu_weights(self, grid, cgroup_ext, delpro, mstart=1, mend=None, into=None):
        """CTUWINI + CTUW for all owned points of `grid` (a ecwam_amd.grid.Grid-like object).
        cgroup_ext: [(npts+1)][NFRE] incl. land row.  Returns dict of reference-shaped weight arrays and
        the (mutated) WLAT/WCOR.  into: the result of an earlier call whose arrays receive the range mstart..mend (the
        second CTUWDRV call of ctuwupdt.F90:241-254); its failure flags are kept."""
        T = self.dtype
        n, nland = grid.nsea, grid.nland
        NANG, NR = self.NANG, self.NFRE_RED
        mend = NR if mend is None else mend
        kxlt = np.ascontiguousarray(grid.kxlt, dtype=np.int32)
        klon = np.ascontiguousarray(grid.klon, dtype=np.int32)
        klat = np.ascontiguousarray(grid.klat, dtype=np.int32)
        kcor = np.ascontiguousarray(grid.kcor, dtype=np.int32)
        wlat = np.array(grid.wlat, dtype=T, order="C")
        wcor = np.array(grid.wcor, dtype=T, order="C")
        cosph = np.ascontiguousarray(grid.cosph, dtype=T)
        sinph = np.ascontiguousarray(grid.sinph, dtype=T)
        zdello = np.ascontiguousarray(grid.zdello, dtype=T)
        cosphm1 = np.ascontiguousarray(grid.cosphm1_ext, dtype=T)
        cg = np.ascontiguousarray(cgroup_ext, dtype=T)
        wlatm1 = np.zeros((n, 2), T)
        wcorm1 = np.zeros((n, 4), T)
        dp = np.zeros((n, 2), T)
        self.lib.ora_ctuwini(C.c_int(n), C.c_int(nland), C.c_int(grid.ngy), self._p(kxlt), self._p(cosph), self._p(cosphm1),
                             self._p(klat), self._p(kcor), self._p(wlat), self._p(wcor), self._p(wlatm1), self._p(wcorm1), self._p(dp))
        if into is not None:
            sumwn, wlonn, wlatn, wcorn, wkpmn = (into[k] for k in ("SUMWN", "WLONN", "WLATN", "WCORN", "WKPMN"))
        else:
            sumwn = np.zeros((n, NANG, NR), T)
            wlonn = np.zeros((n, NANG, NR, 2), T)
            wlatn = np.zeros((n, NANG, NR, 2, 2), T)
            wcorn = np.zeros((n, NANG, NR, 4, 2), T)
            wkpmn = np.zeros((n, NANG, NR, 3), T)
        fail = np.zeros(n, np.int32)
        creal = C.c_float if T == np.float32 else C.c_double
        self.lib.ora_ctuw.restype = C.c_int
        nfail = self.lib.ora_ctuw(C.c_int(n), C.c_int(nland), creal(delpro), C.c_int(mstart), C.c_int(mend), self._p(kxlt),
                                  self._p(zdello), creal(grid.xdella), self._p(cosph), self._p(sinph), self._p(klon), self._p(klat),
                                  self._p(wlat), self._p(wcor), self._p(wlatm1), self._p(wcorm1), self._p(dp), self._p(cg),
                                  self._p(cosphm1), self._p(sumwn), self._p(wlonn), self._p(wlatn), self._p(wcorn), self._p(wkpmn),
                                  self._p(fail))
        if into is not None:
            fail |= into["FAIL"]
            nfail = int(fail.sum())
        return dict(SUMWN=sumwn, WLONN=wlonn, WLATN=wlatn, WCORN=wcorn, WKPMN=wkpmn, WLAT=wlat, WCOR=wcor, NFAIL=nfail, FAIL=fail)

    def ctu_weights_wam(self, grid, cgroup_ext, idelpro, ifrelfmax=0, delpro_lf=None):
        """CTUWUPDT's weight set (ctuwupdt.F90:220-256): one CTUWDRV call over all frequencies, or -- IFRELFMAX > 0 -- DELPRO_LF for
        the fast waves M <= IFRELFMAX and IDELPRO for the rest."""
        if ifrelfmax <= 0:
            return self.ctu_weights(grid, cgroup_ext, float(idelpro))
        w = self.ctu_weights(grid, cgroup_ext, float(delpro_lf), 1, ifrelfmax)
        if ifrelfmax < self.NFRE_RED:
            w = self.ctu_weights(grid, cgroup_ext, float(idelpro), ifrelfmax + 1, self.NFRE_RED, into=w)
        return w

    def propag_wam(self, grid, fl1_ext, w, idelpro, ifrelfmax=0, delpro_lf=None):
        """PROPAG_WAM's advection sequence on one domain (propag_wam.F90:247-313 + the block -> chunk copy :373-386): PROPAGS2 on all
        frequencies, then NSTEP_LF-1 sub-steps on the fast waves.  fl1_ext: [(npts+1)][NANG][NFRE], land row last.  Returns the new
        FL1_EXT (M > NFRE_RED unchanged) and NSTEP_LF."""
        T = self.dtype
        f1 = np.array(fl1_ext, dtype=T, order="C")
        assert f1.shape[0] == grid.nsea + 1
        f3 = np.zeros_like(f1)
        g = self._grid_arrays(grid)
        creal = C.c_float if T == np.float32 else C.c_double
        self.lib.ora_propag_wam.restype = C.c_int
        nstep = self.lib.ora_propag_wam(C.c_int(grid.nsea), self._p(f1), self._p(f3), self._p(g["klon"]), self._p(g["klat"]),
                                        self._p(g["kcor"]), self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]),
                                        self._p(w["WCORN"]), self._p(w["WKPMN"]), C.c_int(int(ifrelfmax)), C.c_int(int(idelpro)),
                                        creal(float(delpro_lf) if delpro_lf else 0.0))
        return f1, nstep

    def propags2(self, grid, f1, w, nd3s=1, nd3e=None):
        """f1: [(npts+1)][NANG][NFRE] (land row zero). Returns F3 with the same shape (rows >= nsea untouched = 0)."""
        T = self.dtype
        nd3e = self.NFRE_RED if nd3e is None else nd3e
        f1 = np.ascontiguousarray(f1, dtype=T)
        f3 = np.zeros_like(f1)
        klon = np.ascontiguousarray(grid.klon, dtype=np.int32)
        klat = np.ascontiguousarray(grid.klat, dtype=np.int32)
        kcor = np.ascontiguousarray(grid.kcor, dtype=np.int32)
        self.lib.ora_propags2(C.c_int(0), C.c_int(grid.nsea), self._p(f1), self._p(f3), self._p(klon), self._p(klat), self._p(kcor),
                              self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]), self._p(w["WCORN"]), self._p(w["WKPMN"]),
                              C.c_int(nd3s), C.c_int(nd3e))
        return f3

    # ---- IREFRA = 1, 2, 3: GRADI + PROPDOT, CTUWDRV/CTUW with every weight, the general PROPAGS2 branch -------------------------
    def _grid_arrays(self, grid):
        T = self.dtype
        return dict(kxlt=np.ascontiguousarray(grid.kxlt, dtype=np.int32), klon=np.ascontiguousarray(grid.klon, dtype=np.int32),
                    klat=np.ascontiguousarray(grid.klat, dtype=np.int32), kcor=np.ascontiguousarray(grid.kcor, dtype=np.int32),
                    wlat=np.array(grid.wlat, dtype=T, order="C"), wcor=np.array(grid.wcor, dtype=T, order="C"),
                    cosph=np.ascontiguousarray(grid.cosph, dtype=T), sinph=np.ascontiguousarray(grid.sinph, dtype=T),
                    zdello=np.ascontiguousarray(grid.zdello, dtype=T), cosphm1=np.ascontiguousarray(grid.cosphm1_ext, dtype=T))

    def propdot(self, grid, irefra, depth_ext, u_ext, v_ext, wavnum_ext, cgroup_ext, omosnh2kd_ext):
        """THDC, THDD [n][NANG] and SDOT [n][NANG][NFRE_RED] (gradi.F90 + propdot.F90).  *_ext: [(npts+1)] / [(npts+1)][NFRE]
        with the land slot filled as proenvhalo.F90:99-106 does."""
        T = self.dtype
        n, nland = grid.nsea, grid.nland
        g = self._grid_arrays(grid)
        creal = C.c_float if T == np.float32 else C.c_double
        a = [np.ascontiguousarray(x, dtype=T) for x in (depth_ext, u_ext, v_ext, wavnum_ext, cgroup_ext, omosnh2kd_ext)]
        thdc = np.zeros((n, self.NANG), T)
        thdd = np.zeros((n, self.NANG), T)
        sdot = np.zeros((n, self.NANG, self.NFRE_RED), T)
        self.lib.ora_propdot(C.c_int(n), C.c_int(nland), C.c_int(irefra), self._p(g["kxlt"]), self._p(g["klon"]), self._p(g["klat"]),
                             self._p(g["wlat"]), self._p(g["zdello"]), creal(grid.xdella), self._p(g["cosph"]), self._p(g["cosphm1"]),
                             *[self._p(x) for x in a], self._p(thdc), self._p(thdd), self._p(sdot))
        return dict(THDC=thdc, THDD=thdd, SDOT=sdot)

    def ctu_weights_gen(self, grid, irefra, cgroup_ext, omosnh2kd_ext, u_ext, v_ext, dot, delpro, llcflcuroff=True, mstart=1, mend=None):
        """CTUWINI + CTUWDRV/CTUW with all weights for IREFRA = 0..3 (`dot` = result of propdot)."""
        T = self.dtype
        n, nland = grid.nsea, grid.nland
        NANG, NR = self.NANG, self.NFRE_RED
        mend = NR if mend is None else mend
        g = self._grid_arrays(grid)
        wlatm1, wcorm1, dp = np.zeros((n, 2), T), np.zeros((n, 4), T), np.zeros((n, 2), T)
        self.lib.ora_ctuwini(C.c_int(n), C.c_int(nland), C.c_int(grid.ngy), self._p(g["kxlt"]), self._p(g["cosph"]),
                             self._p(g["cosphm1"]), self._p(g["klat"]), self._p(g["kcor"]), self._p(g["wlat"]), self._p(g["wcor"]),
                             self._p(wlatm1), self._p(wcorm1), self._p(dp))
        w = dict(SUMWN=np.zeros((n, NANG, NR), T), WLONN=np.zeros((n, NANG, NR, 2), T), WLATN=np.zeros((n, NANG, NR, 2, 2), T),
                 WCORN=np.zeros((n, NANG, NR, 4, 2), T), WKPMN=np.zeros((n, NANG, NR, 3), T), WMPMN=np.zeros((n, NANG, NR, 3), T))
        fail = np.zeros(n, np.int32)
        curmask = np.ones(n, T)
        creal = C.c_float if T == np.float32 else C.c_double
        a = [np.ascontiguousarray(x, dtype=T) for x in (cgroup_ext, omosnh2kd_ext)]
        uv = [np.ascontiguousarray(x, dtype=T) for x in (u_ext, v_ext)]
        self.lib.ora_ctuw_gen.restype = C.c_int
        nfail = self.lib.ora_ctuw_gen(C.c_int(n), C.c_int(irefra), C.c_int(int(llcflcuroff)), creal(delpro), C.c_int(mstart),
                                      C.c_int(mend), self._p(g["kxlt"]), self._p(g["zdello"]), creal(grid.xdella), self._p(g["cosph"]),
                                      self._p(g["sinph"]), self._p(g["klon"]), self._p(g["klat"]), self._p(g["wlat"]), self._p(g["wcor"]),
                                      self._p(wlatm1), self._p(wcorm1), self._p(dp), self._p(a[0]), self._p(a[1]), self._p(g["cosphm1"]),
                                      self._p(uv[0]), self._p(uv[1]), self._p(dot["THDC"]), self._p(dot["THDD"]), self._p(dot["SDOT"]),
                                      self._p(curmask), self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]), self._p(w["WCORN"]),
                                      self._p(w["WKPMN"]), self._p(w["WMPMN"]), self._p(fail))
        w.update(WLAT=g["wlat"], WCOR=g["wcor"], NFAIL=nfail, FAIL=fail, CURMASK=curmask)
        return w

    def propags2_gen(self, grid, f1, w, nd3s=1, nd3e=None):
        """propags2.F90:124-192 (the IREFRA = 2, 3 branch) with the weights of ctu_weights_gen."""
        T = self.dtype
        nd3e = self.NFRE_RED if nd3e is None else nd3e
        f1 = np.ascontiguousarray(f1, dtype=T)
        f3 = np.zeros_like(f1)
        g = self._grid_arrays(grid)
        self.lib.ora_propags2_gen(C.c_int(grid.nsea), self._p(f1), self._p(f3), self._p(g["klon"]), self._p(g["klat"]), self._p(g["kcor"]),
                                  self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]), self._p(w["WCORN"]), self._p(w["WKPMN"]),
                                  self._p(w["WMPMN"]), C.c_int(nd3s), C.c_int(nd3e))
        return f3

    def pack_w8(self, n, w):
        """W8[ij][8][NANG][NFRE_RED]: the eight weights PROPAGS2 reads (ora_pack_w8), for ora_propags2_w8."""
        w8 = np.zeros((n, 8, self.NANG, self.NFRE_RED), self.dtype)
        self.lib.ora_pack_w8(C.c_int(n), self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]), self._p(w["WCORN"]),
                             self._p(w["WKPMN"]), self._p(w8))
        return w8

    def propags2_w8(self, grid, f1, w8, nd3s=1, nd3e=None):
        """PROPAGS2 through the packed weights: same result as propags2()."""
        g = self._grid_arrays(grid)
        n = grid.nsea
        f1 = np.ascontiguousarray(f1, dtype=self.dtype)
        f3 = np.zeros_like(f1)
        self.lib.ora_propags2_w8(C.c_int(0), C.c_int(n), self._p(f1), self._p(f3), self._p(g["klon"]), self._p(g["klat"]), self._p(g["kcor"]),
                                 self._p(w8), C.c_int(nd3s), C.c_int(self.NFRE_RED if nd3e is None else nd3e))
        return f3

    def timed_steps(self, grid, fl, w, props, env, ff, intf, max_steps=50, target_s=15.0, blocked=True):
        """bench.py's cpu_baseline leg: full steps (PROPAGS2 + IMPLSCH) in place on prepared arrays; returns (steps, seconds)
        of the C calls alone (no Python-side copies inside the timed region).  blocked: IMPLSCH through the NPROMA-blocked
        variant (ora_implsch_blk) where it covers the configuration (self.implsch_kind says which ran)."""
        import time

        T = self.dtype
        n = grid.nsea
        f1 = np.ascontiguousarray(fl, dtype=T)
        f3 = np.zeros_like(f1)
        g = self._grid_arrays(grid)
        a = [np.ascontiguousarray(props[k], dtype=T) for k in ("WAVNUM", "CGROUP", "CINV", "XK2CG", "STOKFAC")]
        env = np.ascontiguousarray(env, dtype=T)
        ff = np.array(ff, dtype=T, order="C")
        intf = np.array(intf, dtype=T, order="C")
        xllws = np.zeros((n, self.NANG, self.NFRE), T)
        mij = np.zeros(n, np.int32)
        self.lib.ora_set_ibrmem(None)
        w8 = None
        if blocked:        # the eight weights the stencil reads as contiguous streams (the reference's arrays have IJ fastest); set-up, untimed
            w8 = self.pack_w8(n, w)
        steps, t0 = 0, time.perf_counter()
        self.t_propags2 = self.t_implsch = 0.0      # seconds inside each of the two C calls (separate rates)
        while True:
            ta = time.perf_counter()
            if w8 is not None:
                self.lib.ora_propags2_w8(C.c_int(0), C.c_int(n), self._p(f1), self._p(f3), self._p(g["klon"]), self._p(g["klat"]),
                                         self._p(g["kcor"]), self._p(w8), C.c_int(1), C.c_int(self.NFRE_RED))
            else:
                self.lib.ora_propags2(C.c_int(0), C.c_int(n), self._p(f1), self._p(f3), self._p(g["klon"]), self._p(g["klat"]),
                                      self._p(g["kcor"]), self._p(w["SUMWN"]), self._p(w["WLONN"]), self._p(w["WLATN"]), self._p(w["WCORN"]),
                                      self._p(w["WKPMN"]), C.c_int(1), C.c_int(self.NFRE_RED))
            tb = time.perf_counter()
            rc = -1
            if blocked:
                rc = self.lib.ora_implsch_blk(C.c_int(n), self._p(f3), *(self._p(x) for x in a), self._p(env), self._p(ff), self._p(intf),
                                              self._p(mij), self._p(xllws))
            self.implsch_kind = "nproma-blocked" if rc != -1 else "point by point"
            if rc == -1:
                rc = self.lib.ora_implsch_w2n(C.c_int(n), self._p(f3), *(self._p(x) for x in a), self._p(env), self._p(ff), self._p(intf),
                                              self._p(mij), self._p(xllws), None, None)
            tc = time.perf_counter()
            self.t_propags2 += tb - ta
            self.t_implsch += tc - tb
            if rc:
                raise RuntimeError(f"ora_implsch abort branch rc={rc}")
            f1, f3 = f3, f1          # the land row of both buffers is zero
            steps += 1
            el = time.perf_counter() - t0
            if el > target_s or steps >= max_steps:
                return steps, el
