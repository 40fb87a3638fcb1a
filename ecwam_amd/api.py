"""Host-side mirror of the reference's hot-path interfaces over the C ABI.

`HipContext` owns one `ecwam_hip_ctx`; its methods carry the reference routine names
(PROPAGS2, CTUW, IMPLSCH, NEWWIND) and take torch CUDA tensors purely as device-memory handles
(`data_ptr()` + the current HIP stream).  Shapes, dtypes and index ranges are validated on the host
before any launch, so that a wrong operand cannot reach a kernel.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as _lib
from .tables import Tables

NFF, NINTF, NWPR = 16, 16, 5


class EcwamHipError(RuntimeError):
    pass


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class HipContext:
    def __init__(self, tables: Tables, device: int = 0):
        self.lib = _lib.load()
        self.t = tables
        self.dtype = torch.float32 if tables.dtype == np.float32 else torch.float64
        self.real_bytes = 4 if tables.dtype == np.float32 else 8
        self.NANG, self.NFRE, self.NR = tables.cfg.nang, tables.cfg.nfre, tables.cfg.nfre_red
        self.N = self.NANG * self.NFRE
        self.device = torch.device("cuda", device)
        params = _lib.make_params(tables)
        tp, keep = _lib.make_tables(tables)
        self._h = C.c_void_p()
        rc = self.lib.ecwam_hip_create(C.byref(params), C.byref(tp), self.real_bytes, device, C.byref(self._h))
        del keep
        self._chk(rc)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ecwam_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise EcwamHipError(self.lib.ecwam_hip_last_error().decode())

    # -- validation helpers
    def _real(self, t: torch.Tensor, shape, name: str) -> int:
        if not (t.is_cuda and t.dtype == self.dtype and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
            raise ValueError(f"{name}: expected contiguous {self.dtype} cuda tensor of shape {tuple(shape)}, got "
                             f"{t.dtype} {tuple(t.shape)} cuda={t.is_cuda} contiguous={t.is_contiguous()}")
        return t.data_ptr()

    @staticmethod
    def _int(t: torch.Tensor, shape, name: str) -> int:
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
            raise ValueError(f"{name}: expected contiguous int32 cuda tensor of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
        return t.data_ptr()

    # -- LSUBGRID: OBS[n][8][NFRE] = OBSLAT(:,M,1:2), OBSLON(:,M,1:2), OBSCOR(:,M,1:4); the tensor is kept alive here
    @property
    def has_obstructions(self) -> bool:
        return getattr(self, "_obs", None) is not None

    def set_obstructions(self, obs) -> None:
        if obs is None:
            self._obs = None
            self._chk(self.lib.ecwam_hip_set_obstructions(self._h, None, 0))
            return
        n = obs.shape[0]
        self._obs = obs
        self._chk(self.lib.ecwam_hip_set_obstructions(self._h, self._real(obs, (n, 8, self.NFRE), "OBS"), n))

    # -- PROPAGS2(F1,F3,NINF,NSUP,KIJS,KIJL,NANG,ND3SF1,ND3EF1,ND3S,ND3E)  (propags2.F90:10)
    def propags2(self, f1, f3, klon, klat, kcor, w, kijs, kijl, nd3s=1, nd3e=None, copy_rest=True, check_indices=False):
        nd3e = self.NR if nd3e is None else nd3e
        nrow = f1.shape[0]
        n = klon.shape[0]
        p1 = self._real(f1, (nrow, self.NANG, self.NFRE), "F1")
        p3 = self._real(f3, (nrow, self.NANG, self.NFRE), "F3")
        if not (0 <= kijs <= kijl <= n):
            raise ValueError("PROPAGS2: KIJS/KIJL outside the neighbour tables")
        pk = self._int(klon, (n, 2), "KLON"), self._int(klat, (n, 2, 2), "KLAT"), self._int(kcor, (n, 4, 2), "KCOR")
        pw = self._real(w, (n, 8, self.NANG * self.NR), "W")
        if check_indices:
            for a, nm in ((klon, "KLON"), (klat, "KLAT"), (kcor, "KCOR")):
                lo, hi = int(a.min()), int(a.max())
                if lo < 0 or hi >= nrow:
                    raise ValueError(f"PROPAGS2: {nm} index out of range [{lo},{hi}] for {nrow} spectra")
        self._chk(self.lib.ecwam_hip_propags2(self._h, p1, p3, *pk, pw, kijs, kijl, nd3s, nd3e, int(copy_rest), _stream_ptr()))

    # -- CTUWINI + CTUW (ctuwupdt.F90:204-238)
    def ctuw(self, grid_dev: dict, cgroup_ext, w, cflfail, delpro: float, mstart=1, mend=None):
        mend = self.NR if mend is None else mend
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = cgroup_ext.shape[0]
        if nland >= nrow:
            raise ValueError("CTUW: CGROUP_EXT must include the land row")
        args = [self._int(g["kxlt"], (n,), "KXLT"), self._real(g["zdello"], (ngy,), "ZDELLO"), float(g["xdella"]),
                self._real(g["cosph"], (ngy,), "COSPH"), self._real(g["sinph"], (ngy,), "SINPH"),
                self._int(g["klon"], (n, 2), "KLON"), self._int(g["klat"], (n, 2, 2), "KLAT"), self._int(g["kcor"], (n, 4, 2), "KCOR"),
                self._real(g["wlat"], (n, 2), "WLAT"), self._real(g["wcor"], (n, 4), "WCOR"),
                self._real(cgroup_ext, (nrow, self.NFRE), "CGROUP_EXT"), self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"),
                None if w is None else self._real(w, (n, 8, self.NANG * self.NR), "W"), self._int(cflfail, (n,), "CFLFAIL")]
        self._chk(self.lib.ecwam_hip_ctuw(self._h, n, nland, ngy, float(delpro), mstart, mend, *args, _stream_ptr()))

    # -- PROPAGS2 with on-the-fly CTU weights (no W array): same result as ctuw() + propags2()
    def propags2_otf(self, f1, f3, grid_dev: dict, cgroup_ext, delpro: float, kijs, kijl, nd3s=1, nd3e=None, copy_rest=True,
                     order=None, ifrelfmax: int = 0, delpro_lf: float | None = None, gout=None, tiles2d: bool = False, gin=None):
        """ifrelfmax > 0: frequencies 1..ifrelfmax advance with delpro_lf, the others with delpro, in the same pass.
        gout: optional compact buffer [nrow][NANG][w] that also receives the first w advected frequencies.
        gin: optional compact buffer [nrow][NANG][w] the first w frequencies are READ from (with full rows f1).
        f1 and / or f3 may themselves be compact buffers [nrow][NANG][w] (a fast-wave sub-step)."""
        nd3e = self.NR if nd3e is None else nd3e
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = f1.shape[0]
        if not (0 <= kijs <= kijl) or (order is None and kijl > n) or nland >= nrow or cgroup_ext.shape[0] != nrow:
            raise ValueError("PROPAGS2: KIJS/KIJL outside the neighbour tables, or F1 / CGROUP_EXT without the land row")
        po = None if order is None else self._int(order, (order.shape[0],), "ORDER")
        if order is not None and order.shape[0] < kijl:
            raise ValueError("PROPAGS2: ORDER shorter than KIJL")
        if tiles2d and order is None:
            raise ValueError("PROPAGS2: 2-D tiles need the order of decomp.tile2d_order")
        dlf = float(delpro if delpro_lf is None else delpro_lf)
        in_nfre = int(f1.shape[2])          # NFRE, or the width of a compact fast-wave buffer [nrow][NANG][in_nfre]
        out_nfre = int(f3.shape[2])
        args = [self._real(f1, (nrow, self.NANG, in_nfre), "F1"), self._real(f3, (f3.shape[0], self.NANG, out_nfre), "F3"), n, ngy,
                float(delpro), dlf, int(ifrelfmax), 0 if in_nfre == self.NFRE else in_nfre,
                None if gin is None else self._real(gin, (gin.shape[0], self.NANG, gin.shape[2]), "GIN"),
                0 if gin is None else int(gin.shape[2]), 0 if out_nfre == self.NFRE else out_nfre,
                None if gout is None else self._real(gout, (gout.shape[0], self.NANG, gout.shape[2]), "GOUT"),
                0 if gout is None else int(gout.shape[2]), self._int(g["kxlt"], (n,), "KXLT"), self._real(g["zdello"], (ngy,), "ZDELLO"),
                float(g["xdella"]), self._real(g["cosph"], (ngy,), "COSPH"), self._real(g["sinph"], (ngy,), "SINPH"),
                self._int(g["klon"], (n, 2), "KLON"), self._int(g["klat"], (n, 2, 2), "KLAT"), self._int(g["kcor"], (n, 4, 2), "KCOR"),
                self._real(g["wlat"], (n, 2), "WLAT"), self._real(g["wcor"], (n, 4), "WCOR"),
                self._real(cgroup_ext, (nrow, self.NFRE), "CGROUP_EXT"), self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"), po]
        self._chk(self.lib.ecwam_hip_propags2_otf_fast(self._h, *args, kijs, kijl, nd3s, nd3e, int(bool(copy_rest)) | (4 if tiles2d else 0), _stream_ptr()))

    def set_fastwave_copy(self, g) -> None:
        """g: compact rows [nrow][NANG][w] IMPLSCH / NOSOURCE also write the first w frequencies of their result to (None: off)."""
        if g is None:
            self._chk(self.lib.ecwam_hip_set_fastwave_copy(self._h, None, 0))
        else:
            self._chk(self.lib.ecwam_hip_set_fastwave_copy(self._h, self._real(g, (g.shape[0], self.NANG, g.shape[2]), "G"), int(g.shape[2])))

    # -- FL1_EXT(:,:,M1:M2) <- FL3_EXT between the fast-wave sub-steps (propag_wam.F90:287-291)
    def copy_freq_range(self, src, dst, n, m_first, m_last):
        """dst may be a compact buffer [nrow][NANG][w] with w >= m_last."""
        shape = (src.shape[0], self.NANG, self.NFRE)
        w = int(dst.shape[2])
        if n > src.shape[0] or n > dst.shape[0] or w < m_last:
            raise ValueError("copy_freq_range: shapes")
        self._chk(self.lib.ecwam_hip_copy_freq_range(self._h, self._real(src, shape, "SRC"), self._real(dst, (dst.shape[0], self.NANG, w), "DST"),
                                                     n, m_first, m_last, 0 if w == self.NFRE else w, _stream_ptr()))

    # -- refraction (IREFRA = 1, 2, 3): GRADI + PROPDOT per point, CTUWDRV checks, PROPAGS2 with all weights on the fly
    def _geom(self, g, n, ngy):
        return [self._int(g["kxlt"], (n,), "KXLT"), self._real(g["zdello"], (ngy,), "ZDELLO"), float(g["xdella"]),
                self._real(g["cosph"], (ngy,), "COSPH"), self._real(g["sinph"], (ngy,), "SINPH"),
                self._int(g["klon"], (n, 2), "KLON"), self._int(g["klat"], (n, 2, 2), "KLAT"), self._int(g["kcor"], (n, 4, 2), "KCOR"),
                self._real(g["wlat"], (n, 2), "WLAT"), self._real(g["wcor"], (n, 4), "WCOR")]

    def propdot(self, grid_dev: dict, depth_ext, u_ext, v_ext, refr):
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = depth_ext.shape[0]
        if nland >= nrow:
            raise ValueError("PROPDOT: the *_EXT arrays must include the land row")
        self._chk(self.lib.ecwam_hip_propdot(
            self._h, n, nland, self._int(g["kxlt"], (n,), "KXLT"), self._real(g["zdello"], (ngy,), "ZDELLO"), float(g["xdella"]),
            self._real(g["cosph"], (ngy,), "COSPH"), self._int(g["klon"], (n, 2), "KLON"), self._int(g["klat"], (n, 2, 2), "KLAT"),
            self._real(g["wlat"], (n, 2), "WLAT"), self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"),
            self._real(depth_ext, (nrow,), "DEPTH_EXT"), self._real(u_ext, (nrow,), "U_EXT"), self._real(v_ext, (nrow,), "V_EXT"),
            self._real(refr, (n, 2 * self.NANG + 5), "REFR"), _stream_ptr()))

    def ctuw_refra(self, grid_dev: dict, cgroup_ext, omosnh2kd_ext, wavnum_ext, refr, cflfail, delpro: float, mstart=1, mend=None,
                   llcflcuroff=True, frange=0):
        mend = self.NR if mend is None else mend
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = cgroup_ext.shape[0]
        if nland >= nrow:
            raise ValueError("CTUW: CGROUP_EXT must include the land row")
        ext = [self._real(a, (nrow, self.NFRE), nm) for a, nm in ((cgroup_ext, "CGROUP_EXT"), (omosnh2kd_ext, "OMOSNH2KD_EXT"),
                                                                 (wavnum_ext, "WAVNUM_EXT"))]
        self._chk(self.lib.ecwam_hip_ctuw_refra(self._h, n, nland, ngy, float(delpro), mstart, mend, *self._geom(g, n, ngy), *ext,
                                                self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"),
                                                self._real(refr, (n, 2 * self.NANG + 5), "REFR"), int(llcflcuroff), int(frange),
                                                self._int(cflfail, (n,), "CFLFAIL"), _stream_ptr()))

    def propags2_refra(self, f1, f3, grid_dev: dict, cgroup_ext, omosnh2kd_ext, wavnum_ext, refr, delpro: float, kijs, kijl, nd3s=1,
                       nd3e=None, copy_rest=True, frange=0):
        nd3e = self.NR if nd3e is None else nd3e
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = f1.shape[0]
        if not (0 <= kijs <= kijl <= n) or nland >= nrow or cgroup_ext.shape[0] != nrow:
            raise ValueError("PROPAGS2: KIJS/KIJL outside the neighbour tables, or F1 / CGROUP_EXT without the land row")
        ext = [self._real(a, (nrow, self.NFRE), nm) for a, nm in ((cgroup_ext, "CGROUP_EXT"), (omosnh2kd_ext, "OMOSNH2KD_EXT"),
                                                                 (wavnum_ext, "WAVNUM_EXT"))]
        self._chk(self.lib.ecwam_hip_propags2_refra(
            self._h, self._real(f1, (nrow, self.NANG, self.NFRE), "F1"), self._real(f3, (nrow, self.NANG, self.NFRE), "F3"), n, ngy,
            float(delpro), *self._geom(g, n, ngy), *ext, self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"),
            self._real(refr, (n, 2 * self.NANG + 5), "REFR"), int(frange), kijs, kijl, nd3s, nd3e, int(copy_rest), _stream_ptr()))

    # -- IMPLSCH (implsch.F90:10-23)
    def implsch(self, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, dbg=None, wam2nemo=None):
        nrow = fl1.shape[0]
        if not (0 <= kijs <= kijl <= min(nrow, wvprpt.shape[0], ff.shape[0], intf.shape[0], mij.shape[0], xllws.shape[0])):
            raise ValueError("IMPLSCH: KIJS/KIJL outside the operands")
        a = [self._real(fl1, (nrow, self.NANG, self.NFRE), "FL1"), self._real(wvprpt, (wvprpt.shape[0], NWPR, self.NFRE), "WVPRPT"),
             self._real(ff, (ff.shape[0], NFF), "FF"), self._real(intf, (intf.shape[0], NINTF), "INTF"),
             self._int(mij, (mij.shape[0],), "MIJ"), self._real(xllws, (xllws.shape[0], self.NANG, self.NFRE), "XLLWS")]
        pd = None if dbg is None else self._real(dbg, (nrow, 32), "DBG")
        pw = None
        if wam2nemo is not None:
            if not (wam2nemo.is_cuda and wam2nemo.dtype == torch.float64 and wam2nemo.is_contiguous() and wam2nemo.dim() == 2
                    and wam2nemo.shape[1] == 13 and wam2nemo.shape[0] >= kijl):
                raise ValueError("WAM2NEMO: expected contiguous float64 cuda tensor [npts >= KIJL][13]")
            pw = wam2nemo.data_ptr()
        self._chk(self.lib.ecwam_hip_implsch(self._h, kijs, kijl, *a, pw, pd, _stream_ptr()))

    # -- the one-kernel step: PROPAGS2 inside IMPLSCH's tile load (ecwam_hip_propags2_implsch)
    def fused_supported(self, fast_waves: bool = False, obstructions: bool = False) -> bool:
        """The one-kernel step covers the context (and, if asked, its forms with fast-wave sub-steps / sub-grid obstructions)."""
        mask = int(self.lib.ecwam_hip_propags2_implsch_supported(self._h))
        need = 1 | (2 if fast_waves else 0) | (4 if obstructions else 0)
        return (mask & need) == need

    def propags2_implsch(self, f1, f3, grid_dev: dict, cgroup_ext, delpro: float, kijs, kijl, wvprpt, ff, intf, mij, xllws, nd3s=1, nd3e=None,
                         wam2nemo=None, flags: int = 0, ifrelfmax: int = 0, delpro_lf: float | None = None, gin=None):
        """Rows [kijs, kijl): advect from the rows of f1 (read only) and integrate the source terms; the new spectrum goes to the rows of f3.
        Bit for bit propags2_otf(f1 -> f3) followed by implsch(f3).  ifrelfmax > 0 with gin: frequencies 1..ifrelfmax advance with delpro_lf
        from the compact rows gin [nrow][NANG][w] (the last fast-wave sub-step), the others with delpro from f1."""
        nd3e = self.NR if nd3e is None else nd3e
        g = grid_dev
        n, nland, ngy = g["n"], g["nland"], g["ngy"]
        nrow = f1.shape[0]
        if not (0 <= kijs <= kijl <= n) or nland >= nrow or cgroup_ext.shape[0] != nrow or f3.shape[0] < kijl:
            raise ValueError("PROPAGS2 + IMPLSCH: KIJS/KIJL outside the neighbour tables, or F1 / CGROUP_EXT without the land row")
        if not (kijl <= min(wvprpt.shape[0], ff.shape[0], intf.shape[0], mij.shape[0], xllws.shape[0])):
            raise ValueError("PROPAGS2 + IMPLSCH: KIJL outside the operands")
        pw = None
        if wam2nemo is not None:
            if not (wam2nemo.is_cuda and wam2nemo.dtype == torch.float64 and wam2nemo.is_contiguous() and wam2nemo.dim() == 2
                    and wam2nemo.shape[1] == 13 and wam2nemo.shape[0] >= kijl):
                raise ValueError("WAM2NEMO: expected contiguous float64 cuda tensor [npts >= KIJL][13]")
            pw = wam2nemo.data_ptr()
        args = [self._real(f1, (nrow, self.NANG, self.NFRE), "F1"), self._real(f3, (f3.shape[0], self.NANG, self.NFRE), "F3"), n, ngy, float(delpro),
                self._int(g["kxlt"], (n,), "KXLT"), self._real(g["zdello"], (ngy,), "ZDELLO"), float(g["xdella"]),
                self._real(g["cosph"], (ngy,), "COSPH"), self._real(g["sinph"], (ngy,), "SINPH"),
                self._int(g["klon"], (n, 2), "KLON"), self._int(g["klat"], (n, 2, 2), "KLAT"), self._int(g["kcor"], (n, 4, 2), "KCOR"),
                self._real(g["wlat"], (n, 2), "WLAT"), self._real(g["wcor"], (n, 4), "WCOR"),
                self._real(cgroup_ext, (nrow, self.NFRE), "CGROUP_EXT"), self._real(g["cosphm1_ext"], (nrow,), "COSPHM1_EXT"), kijs, kijl, nd3s, nd3e,
                self._real(wvprpt, (wvprpt.shape[0], NWPR, self.NFRE), "WVPRPT"), self._real(ff, (ff.shape[0], NFF), "FF"),
                self._real(intf, (intf.shape[0], NINTF), "INTF"), self._int(mij, (mij.shape[0],), "MIJ"),
                self._real(xllws, (xllws.shape[0], self.NANG, self.NFRE), "XLLWS"), pw,
                float(delpro if delpro_lf is None else delpro_lf), int(ifrelfmax),
                None if gin is None else self._real(gin, (gin.shape[0], self.NANG, gin.shape[2]), "GIN"), 0 if gin is None else int(gin.shape[2]),
                int(flags), _stream_ptr()]
        self._chk(self.lib.ecwam_hip_propags2_implsch(self._h, *args))

    def implsch_reserve(self, npts: int) -> None:
        """Size the per-point scalar rows of the IMPLSCH kernels once, outside the time loop: ecwam_hip_implsch_reserve."""
        self._chk(self.lib.ecwam_hip_implsch_reserve(self._h, int(npts)))

    def implsch_generation_used(self) -> int:
        """Kernel generation the last implsch() call launched: 4 (k_implsch4) since the one-point-per-wavefront kernel left the product."""
        return int(self.lib.ecwam_hip_implsch_generation_used(self._h))

    def device_tables(self) -> int:
        """Address of the device copy of the module tables (ecwam_hip_device_tables): diagnostics, the tests' second IMPLSCH implementation."""
        return int(self.lib.ecwam_hip_device_tables(self._h) or 0)

    def outbs(self, kijs, kijl, fl1, out, zmiss: float = -999.0):
        nrow = fl1.shape[0]
        if not (0 <= kijs <= kijl <= min(nrow, out.shape[0])):
            raise ValueError("OUTBS: KIJS/KIJL outside the operands")
        self._chk(self.lib.ecwam_hip_outbs(self._h, kijs, kijl, self._real(fl1, (nrow, self.NANG, self.NFRE), "FL1"), float(zmiss),
                                           self._real(out, (out.shape[0], 5), "OUT"), _stream_ptr()))

    def outwnorm(self, field, column: int, n: int, zmiss: float = -999.0):
        """(average, minimum, maximum, count) of field[:n, column] over the values != zmiss."""
        if not (field.is_cuda and field.dtype == self.dtype and field.is_contiguous() and field.dim() == 2 and n <= field.shape[0]):
            raise ValueError("OUTWNORM: expected a contiguous 2-D cuda tensor in the working precision")
        import ctypes as C
        res = (C.c_double * 4)()
        ptr = field.data_ptr() + column * field.element_size()
        self._chk(self.lib.ecwam_hip_outwnorm(self._h, ptr, field.shape[1], n, float(zmiss), res, _stream_ptr()))
        return tuple(res)

    # -- NEWWIND (newwind.F90:126-161)
    def newwind(self, ff, ff_next, icode_wnd: int | None = None):
        """icode_wnd: ICODE_CPL of a coupled run (newwind.F90:120-124); default ICODE of the parameters."""
        n = ff.shape[0]
        a = (self._h, n, self._real(ff, (n, NFF), "FF"), self._real(ff_next, (n, NFF), "FF_NEXT"))
        if icode_wnd is None:
            self._chk(self.lib.ecwam_hip_newwind(*a, _stream_ptr()))
        else:
            self._chk(self.lib.ecwam_hip_newwind_icode(*a, int(icode_wnd), _stream_ptr()))

    def nosource(self, kijs, kijl, fl1, mij, xllws):
        """LLSOURCE = F (wamintgr.F90:152-160): FL1 = MAX(FL1, EPSMIN), MIJ = NFRE, XLLWS = 0 on rows [kijs, kijl).
        fl1 = None: a call before the next source-term date (wamintgr.F90:178-186): MIJ and XLLWS only."""
        nrow = xllws.shape[0] if fl1 is None else fl1.shape[0]
        if not (0 <= kijs <= kijl <= min(nrow, mij.shape[0], xllws.shape[0])):
            raise ValueError("NOSOURCE: KIJS/KIJL outside the operands")
        self._chk(self.lib.ecwam_hip_nosource(self._h, kijs, kijl, None if fl1 is None else self._real(fl1, (nrow, self.NANG, self.NFRE), "FL1"),
                                              self._int(mij, (mij.shape[0],), "MIJ"),
                                              self._real(xllws, (xllws.shape[0], self.NANG, self.NFRE), "XLLWS"), _stream_ptr()))

    # -- layout conversion (propag_wam.F90:124-137, 373-400)
    def chunks_to_points(self, chunked, points, nproma, nchnk, npts, n2, n3):
        pc = self._real(chunked, (nchnk, n3, n2, nproma), "chunked")
        pp = self._real(points, (points.shape[0], n2, n3), "points")
        if points.shape[0] < npts:
            raise ValueError("points buffer too small")
        self._chk(self.lib.ecwam_hip_chunks_to_points(self._h, pc, pp, nproma, nchnk, npts, n2, n3, _stream_ptr()))

    def points_to_chunks(self, points, chunked, nproma, nchnk, npts, n2, n3):
        pc = self._real(chunked, (nchnk, n3, n2, nproma), "chunked")
        pp = self._real(points, (points.shape[0], n2, n3), "points")
        if points.shape[0] < npts:
            raise ValueError("points buffer too small")
        self._chk(self.lib.ecwam_hip_points_to_chunks(self._h, pp, pc, nproma, nchnk, npts, n2, n3, _stream_ptr()))

    # -- halo pack / unpack (mpexchng.F90:124-138, 217-231)
    def pack_rows(self, fl, idx, buf):
        n = idx.shape[0]
        if n == 0:
            return
        self._chk(self.lib.ecwam_hip_pack_rows(self._h, self._real(fl, (fl.shape[0], self.NANG, self.NFRE), "FL"),
                                               self._int(idx, (n,), "IDX"), n, self._real(buf, (n, self.NANG, self.NFRE), "BUF"),
                                               _stream_ptr()))

    def unpack_rows(self, buf, fl, dst0):
        n = buf.shape[0]
        if n == 0:
            return
        if dst0 < 0 or dst0 + n > fl.shape[0]:
            raise ValueError("unpack_rows: destination range outside FL")
        self._chk(self.lib.ecwam_hip_unpack_rows(self._h, self._real(buf, (n, self.NANG, self.NFRE), "BUF"), n,
                                                 self._real(fl, (fl.shape[0], self.NANG, self.NFRE), "FL"), dst0, _stream_ptr()))


    # -- MPEXCHNG inside the library (include/ecwam_hip.h: ecwam_hip_halo_*)
    def halo_setup(self, dom) -> None:
        """dom: decomp.LocalDomain.  Peers in ascending rank order; send lists concatenated in that order."""
        peers = sorted(set(dom.send) | set(dom.recv))
        self._halo_peers = peers
        pa = np.asarray(peers, dtype=np.int32)
        sc = np.asarray([len(dom.send.get(p, ())) for p in peers], dtype=np.int32)
        si = np.concatenate([np.asarray(dom.send[p], dtype=np.int32) for p in peers if p in dom.send]) if sc.sum() else np.zeros(0, np.int32)
        rd = np.asarray([dom.recv.get(p, (0, 0))[0] for p in peers], dtype=np.int32)
        rc = np.asarray([dom.recv.get(p, (0, 0))[1] for p in peers], dtype=np.int32)
        self._halo_send_cnt, self._halo_recv_cnt = sc, rc
        ptr = lambda a: a.ctypes.data if a.size else None
        self._chk(self.lib.ecwam_hip_halo_setup(self._h, dom.rank, dom.nranks, len(peers), ptr(pa), ptr(sc), ptr(si), ptr(rd), ptr(rc)))

    def comm_unique_id(self) -> bytes:
        import ctypes
        buf = ctypes.create_string_buffer(128)
        self._chk(self.lib.ecwam_hip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes) -> None:
        if len(uid) != 128:
            raise ValueError("RCCL unique id: 128 bytes")
        self._chk(self.lib.ecwam_hip_comm_init(self._h, uid))

    def comm_count(self) -> int:
        """Ranks of the library's RCCL communicator as RCCL reports them (0: none)."""
        k = C.c_int(0)
        self._chk(self.lib.ecwam_hip_comm_count(self._h, C.byref(k)))
        return int(k.value)

    def _rows(self, fl):
        if not (fl.is_cuda and fl.is_contiguous() and fl.dtype == self.dtype and fl.dim() == 3):
            raise ValueError("halo: expected a contiguous device tensor [rows][NANG][M] of the context's precision")
        return int(fl.shape[1] * fl.shape[2])

    def halo_start(self, fl) -> None:
        self._chk(self.lib.ecwam_hip_halo_start(self._h, fl.data_ptr(), self._rows(fl), _stream_ptr()))

    def halo_finish(self) -> None:
        self._chk(self.lib.ecwam_hip_halo_finish(self._h, _stream_ptr()))

    # -- PROENVHALO on the device (proenvhalo.F90:63-107)
    def proenvhalo_pack(self, n, wvprpt, omosnh2kd, depth, ucur, vcur, buffer_ext) -> None:
        self._chk(self.lib.ecwam_hip_proenvhalo_pack(self._h, int(n), wvprpt.data_ptr(), omosnh2kd.data_ptr(), depth.data_ptr(), ucur.data_ptr(),
                                                     vcur.data_ptr(), buffer_ext.data_ptr(), _stream_ptr()))

    def proenvhalo_unpack(self, nrows, buffer_ext, land, wavnum_ext, cgroup_ext, omosnh2kd_ext, depth_ext, u_ext, v_ext) -> None:
        self._chk(self.lib.ecwam_hip_proenvhalo_unpack(self._h, int(nrows), buffer_ext.data_ptr(), land.data_ptr(), wavnum_ext.data_ptr(),
                                                       cgroup_ext.data_ptr(), omosnh2kd_ext.data_ptr(), depth_ext.data_ptr(), u_ext.data_ptr(),
                                                       v_ext.data_ptr(), _stream_ptr()))

    def halo_pack_host(self, fl, host_send) -> None:
        self._chk(self.lib.ecwam_hip_halo_pack_host(self._h, fl.data_ptr(), self._rows(fl), host_send.data_ptr(), _stream_ptr()))

    def halo_unpack_host(self, fl, host_recv) -> None:
        self._chk(self.lib.ecwam_hip_halo_unpack_host(self._h, fl.data_ptr(), self._rows(fl), host_recv.data_ptr(), _stream_ptr()))


def grid_to_device(grid, dtype, device, lo: int = 0, hi: int | None = None, local=None) -> dict:
    """Upload the grid tables of points [lo,hi) (or a decomp.LocalDomain) as the dict `HipContext.ctuw` expects."""
    npdt = np.float32 if dtype == torch.float32 else np.float64
    if local is not None:
        klon, klat, kcor, kxlt = local.klon, local.klat, local.kcor, local.kxlt
        wlat, wcor = grid.wlat[local.lo:local.hi], grid.wcor[local.lo:local.hi]
        n, nland, cosphm1 = local.n, local.nland, local.cosphm1_ext
    else:
        hi = grid.nsea if hi is None else hi
        assert lo == 0 and hi == grid.nsea
        klon, klat, kcor, kxlt = grid.klon, grid.klat, grid.kcor, grid.kxlt
        wlat, wcor, n, nland, cosphm1 = grid.wlat, grid.wcor, grid.nsea, grid.nland, grid.cosphm1_ext

    def ti(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)

    def tr(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=npdt)).to(device)

    return dict(n=n, nland=nland, ngy=grid.ngy, xdella=grid.xdella, kxlt=ti(kxlt), zdello=tr(grid.zdello), cosph=tr(grid.cosph),
                sinph=tr(grid.sinph), klon=ti(klon), klat=ti(klat), kcor=ti(kcor), wlat=tr(wlat), wcor=tr(wcor),
                cosphm1_ext=tr(cosphm1))
