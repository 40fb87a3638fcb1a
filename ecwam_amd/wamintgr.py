"""WAMINTGR on the device: the reference's per-step call sequence (wamintgr.F90:94-146)

    IF (CDATE == CDTPRA) CALL PROPAG_WAM      -> halo exchange (MPEXCHNG) + PROPAGS2 (+ fast-wave sub-steps)
    CALL NEWWIND                              -> forcing hand-over when a new wind field is due
    IF (CDATE >= CDTIMPNEXT) CALL IMPLSCH     -> source-term integration of every owned point

with the state resident in HBM between steps (the GPU variant's behaviour, wamintgr_loki_gpu.F90:99-201).
One instance = one rank = one GPU; ranks own contiguous sea-point ranges (decomp.py) and exchange
halo spectra point-to-point through torch.distributed (backend "nccl" = RCCL over xGMI).
"""
from __future__ import annotations

import numpy as np
import torch

from . import api, decomp, synthetic as syn
from .tables import Config, Tables


class HaloExchange:
    """MPEXCHNG (mpexchng.F90:141-206): pack -> point-to-point exchange with the neighbouring ranks -> halo rows.

    transport "torch": torch.distributed P2P (backend "nccl" = RCCL) on tensors packed by ecwam_hip_pack_rows.
    transport "lib":   the library's own exchange (ecwam_hip_halo_start / _finish: RCCL loaded by the library, its own stream) --
                       what a Fortran host calls; the unique id travels through torch.distributed here, MPI_Bcast there.
    transport "host":  host-staged (ecwam_hip_halo_pack_host / _unpack_host) with the segments exchanged through the process
                       group's CPU backend (gloo): an MPI without device pointers, or several ranks sharing one GPU in tests."""

    def __init__(self, dom: decomp.LocalDomain, device, ctx=None, transport: str = "torch"):
        self.dom = dom
        self.ctx = ctx
        self.transport = transport
        if transport not in ("torch", "lib", "host"):
            raise ValueError("halo transport: torch, lib or host")
        self.send_idx = {p: torch.from_numpy(np.ascontiguousarray(ix)).to(device) for p, ix in dom.send.items()}
        self.bufs = {}
        if transport != "torch" and dom.nranks > 1:
            if ctx is None:
                raise ValueError("library halo transports need the HipContext")
            ctx.halo_setup(dom)
            if transport == "lib":
                import torch.distributed as dist
                uid = torch.zeros(128, dtype=torch.uint8)
                if dom.rank == 0:
                    uid = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8).clone()
                try:                          # a CPU tensor where the group has a CPU backend (gloo, or "cpu:gloo,cuda:nccl") ...
                    dist.broadcast(uid, src=0)
                except RuntimeError:          # ... a device tensor in a pure RCCL group (the same on every rank)
                    t = uid.to(device)
                    dist.broadcast(t, src=0)
                    uid = t.cpu()
                ctx.comm_init(bytes(uid.numpy().tobytes()))

    def start(self, fl: torch.Tensor) -> list:
        """Pack the rows the neighbours need and post every send / receive; returns the pending requests.  Until `finish`
        the caller must leave the halo rows [n, n+nh) of `fl` alone (the receives land there) -- the owned rows may be read."""
        if self.dom.nranks == 1:
            return []
        import torch.distributed as dist

        if self.transport == "lib":
            self.ctx.halo_start(fl)
            return [None]
        if self.transport == "host":
            rl = int(fl.shape[1] * fl.shape[2])
            sc, rc = self.ctx._halo_send_cnt, self.ctx._halo_recv_cnt
            hs = torch.empty((int(sc.sum()), rl), dtype=fl.dtype).pin_memory()
            hr = torch.empty((int(rc.sum()), rl), dtype=fl.dtype).pin_memory()
            self.ctx.halo_pack_host(fl, hs)
            ops, so, ro = [], 0, 0
            for p, ns, nr in zip(self.ctx._halo_peers, sc, rc):
                if ns:
                    ops.append(dist.P2POp(dist.isend, hs[so:so + int(ns)], p))
                if nr:
                    ops.append(dist.P2POp(dist.irecv, hr[ro:ro + int(nr)], p))
                so += int(ns); ro += int(nr)
            reqs = dist.batch_isend_irecv(ops) if ops else []
            return [("host", reqs, fl, hr, hs)]
        ops = []
        for p, ix in sorted(self.send_idx.items()):
            buf = self.bufs.get((p, int(fl.shape[2])))
            if buf is None or buf.dtype != fl.dtype or buf.shape[1:] != fl.shape[1:]:
                buf = self.bufs[(p, int(fl.shape[2]))] = torch.empty((ix.numel(),) + tuple(fl.shape[1:]), dtype=fl.dtype, device=fl.device)
            if fl.is_cuda and self.ctx is not None and fl.shape[2] == self.ctx.NFRE:
                self.ctx.pack_rows(fl, ix, buf)
            else:
                torch.index_select(fl, 0, ix.long(), out=buf)
            ops.append(dist.P2POp(dist.isend, buf, p))
        for p, (dst0, cnt) in sorted(self.dom.recv.items()):
            ops.append(dist.P2POp(dist.irecv, fl[dst0:dst0 + cnt], p))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(self, reqs: list) -> None:
        for r in reqs:
            if r is None:
                self.ctx.halo_finish()
            elif isinstance(r, tuple):
                _, rr, fl, hr, _hs = r
                for q in rr:
                    q.wait()
                self.ctx.halo_unpack_host(fl, hr)
            else:
                r.wait()

    def __call__(self, fl: torch.Tensor) -> None:
        self.finish(self.start(fl))


class Wamintgr:
    """Device-resident WAMINTGR for one rank."""

    def __init__(self, cfg: Config, grid, prec: str = "sp", device: int = 0, rank: int = 0, nranks: int = 1,
                 ifrelfmax: int = 0, delpro_lf: float | None = None, weights: str = "otf", strip_width: int = 0,
                 halo_transport: str = "torch"):
        # weights: "otf" rebuilds the CTU weights inside PROPAGS2 (no W array, default); "stored" keeps the reference's
        # scheme (CTUW once into W[ij][8][NANG*NFRE_RED], PROPAGS2 streams them).  Bit-identical results.
        self.cfg, self.grid, self.prec = cfg, grid, prec
        self.npdt = np.float32 if prec == "sp" else np.float64
        self.t = Tables(cfg, self.npdt)
        self.ctx = api.HipContext(self.t, device)
        self.dev, self.dtype = self.ctx.device, self.ctx.dtype
        self.dom = decomp.local_domain(grid, rank, nranks)
        self.n, self.nrows = self.dom.n, self.dom.nrows
        self.gd = api.grid_to_device(grid, self.dtype, self.dev, local=self.dom)
        self.halo = HaloExchange(self.dom, self.dev, self.ctx, transport=halo_transport)
        self.interior = self.dom.interior()      # rows [a, b) whose stencil reads no halo row
        NANG, NFRE, NR = cfg.nang, cfg.nfre, cfg.nfre_red
        z = dict(dtype=self.dtype, device=self.dev)
        self.fl1 = torch.zeros((self.nrows, NANG, NFRE), **z)
        self.fl3 = torch.zeros((self.nrows, NANG, NFRE), **z)
        if weights not in ("otf", "stored"):
            raise ValueError("weights must be 'otf' or 'stored'")
        self.weights = weights
        self.w = torch.zeros((self.n, 8, NANG * NR), **z) if weights == "stored" else None
        # optional processing order of the advection (longitude strips); measured neutral on MI355X (the 256 MB Infinity Cache
        # already serves the latitude neighbours), kept as an option
        self.order = None
        self.tiles2d = False
        if strip_width > 0 and self.n > 4 * strip_width:
            self.order = torch.from_numpy(decomp.strip_order(grid, self.dom, strip_width)).to(self.dev)
        elif strip_width < 0:      # 2-D tiles of four latitude rows x four longitudes
            self.order = torch.from_numpy(decomp.tile2d_order(grid, self.dom)).to(self.dev)
            self.tiles2d = True
        if self.order is not None and (int(cfg.irefra) or weights == "stored"):
            # the refraction and stored-weight kernels walk the rows in their natural order: a work order (whose padded length would
            # otherwise replace the row range handed to them) does not apply
            self.order, self.tiles2d = None, False
        self.wvprpt = torch.zeros((self.n, api.NWPR, NFRE), **z)
        self.ff = torch.zeros((self.n, api.NFF), **z)
        self.ff_next = None
        self.intf = torch.zeros((self.n, api.NINTF), **z)
        self.mij = torch.zeros(self.n, dtype=torch.int32, device=self.dev)
        self.xllws = torch.zeros((self.n, NANG, NFRE), **z)
        self.ctx.implsch_reserve(self.n)        # no allocation inside the time loop
        self.cflfail = torch.zeros(self.n, dtype=torch.int32, device=self.dev)
        self.ifrelfmax = ifrelfmax
        self.delpro_lf = delpro_lf
        # fast waves between the sub-steps: compact rows [ij][K][LFP] (a frequency sub-range of the full rows would touch every
        # cache line of the spectra; these are 36/LFP times smaller, for the stencil and for the halo exchange)
        self.g1 = self.g2 = None
        self.gfast_valid = False        # g1 holds the first LFP frequencies of the current FL1 (fast waves + the slow ones that fill its last vector)
        # "compact": sub-steps 1 .. NSTEP_LF-1 compact -> compact BEFORE the full pass, which reads the fast waves' last state from the
        # compact buffer and writes complete rows (round 3); "rows": the full pass first, the further sub-steps write the fast-wave slots
        # of the FL3 rows (round 2; every line of FL3 touched once more per sub-step).  Same bits.
        self._fast_mode = "compact"
        if 0 < ifrelfmax < cfg.nfre_red and weights == "otf" and not int(cfg.irefra):
            lfp = min(cfg.nfre, (ifrelfmax + 3) // 4 * 4)
            self.g1 = torch.zeros((self.nrows, NANG, lfp), **z)
            self.g2 = torch.zeros((self.nrows, NANG, lfp), **z)
        self.weights_ready = False
        self.halo_events = None         # a list: propag() appends a (before, after) event pair around every wait for a halo exchange
        # refraction (IREFRA = 1 depth, 2 currents, 3 both): per-point THD/S0/U/V/OMDD/CURMASK instead of the reference's
        # THDD/THDC/SDOT and 21 weight arrays; PROPAGS2 rebuilds every weight on the fly
        self.irefra = int(cfg.irefra)
        self.llcflcuroff = True                 # mpuserin.F90:575
        if self.irefra:
            if weights != "otf":
                raise ValueError("IREFRA != 0 runs with on-the-fly weights only")
            self.refr = torch.zeros((self.n, 2 * NANG + 5), **z)
            self.depth_ext = torch.zeros(self.nrows, **z)
            self.u_ext = torch.zeros(self.nrows, **z)
            self.v_ext = torch.zeros(self.nrows, **z)
            self.omosnh2kd_ext = torch.zeros((self.nrows, NFRE), **z)
            self.wavnum_ext = torch.zeros((self.nrows, NFRE), **z)

    @property
    def fast_mode(self) -> str:
        return self._fast_mode

    @fast_mode.setter
    def fast_mode(self, mode: str) -> None:
        if mode not in ("compact", "rows"):
            raise ValueError("fast_mode: 'compact' or 'rows'")
        self._fast_mode = mode
        self.gfast_valid = False      # the rows mode uses g1 as scratch: whatever it holds no longer describes FL1

    # ---- synthetic initial state; identical for every decomposition
    def init_synthetic(self, seed: int = 12345, chunk: int = 65536, currents: bool | None = None, env_on_device: bool = False) -> None:
        """env_on_device (refraction on a decomposed grid): the extended DEPTH / UCUR / VCUR / OMOSNH2KD / WAVNUM / CGROUP rows are not
        assembled on the host from the global fields but by PROENVHALO on the device -- owned rows + one halo exchange (proenvhalo())."""
        g, d, t = self.grid, self.dom, self.t
        self.gfast_valid = False      # new spectra: the compact fast-wave rows are extracted again by the next propag()
        p = syn.point_params(g.nsea, seed=seed)
        self.params = p
        ext = d.ext_global()
        props_ext_cg = np.zeros((self.nrows, self.cfg.nfre), self.npdt)
        if self.irefra:
            om_ext = np.zeros((self.nrows, self.cfg.nfre), self.npdt)
            wn_ext = np.zeros((self.nrows, self.cfg.nfre), self.npdt)
        for s in range(0, ext.size, chunk):
            sl = ext[s:s + chunk]
            pr = syn.depth_props(p["DEPTH"][sl], t, self.npdt)
            props_ext_cg[s:s + sl.size] = pr["CGROUP"]
            if self.irefra:
                om_ext[s:s + sl.size] = pr["OMOSNH2KD"]
                wn_ext[s:s + sl.size] = pr["WAVNUM"]
            own = sl[(sl >= d.lo) & (sl < d.hi)]
            if own.size:
                a, b = s, s + own.size  # owned rows come first in ext
                wv = np.stack([pr[k][: own.size] for k in ("WAVNUM", "CGROUP", "CINV", "XK2CG", "STOKFAC")], 1)
                self.wvprpt[a:b] = torch.from_numpy(wv).to(self.dev)
                ff = np.zeros((own.size, api.NFF), self.npdt)
                ff[:, :14] = syn.forcing(p, own, t, self.npdt)
                ff[:, 14] = pr["EMAXDPT"][: own.size]
                ff[:, 15] = p["DEPTH"][own]
                self.ff[a:b] = torch.from_numpy(ff).to(self.dev)
                fl = syn.jonswap_spectra(t.FR, t.TH, p["FP"][own], p["THETAQ"][own], self.npdt)
                self.fl1[a:b] = torch.from_numpy(fl).to(self.dev)
        land = syn.depth_props(np.array([float(t.BATHYMAX)]), t, self.npdt)      # WVPRPT_LAND (initdpthflds.F90:85-88)
        props_ext_cg[self.dom.nland] = land["CGROUP"][0]
        self.cgroup_ext = torch.from_numpy(props_ext_cg).to(self.dev)
        if self.irefra:
            # PROENVHALO (proenvhalo.F90:69-106): own + halo values, land slot = (WVPRPT_LAND, BATHYMAX, U = V = 0)
            om_ext[self.dom.nland] = land["OMOSNH2KD"][0]
            wn_ext[self.dom.nland] = land["WAVNUM"][0]
            dep = np.full(self.nrows, float(t.BATHYMAX), self.npdt)
            dep[: ext.size] = p["DEPTH"][ext]
            u = np.zeros(self.nrows, self.npdt)
            v = np.zeros(self.nrows, self.npdt)
            if currents if currents is not None else self.irefra >= 2:
                ug, vg = syn.currents(g)
                u[: ext.size] = ug[ext]
                v[: ext.size] = vg[ext]
            if env_on_device:
                n = self.n
                self.proenvhalo(dep[:n], u[:n], v[:n], om_ext[:n])
            else:
                self.set_environment(dep, u, v, om_ext, wn_ext)
        self.fl1[self.dom.nland].zero_()
        self.fl3[self.dom.nland].zero_()

    def set_obstructions(self, obs_global) -> None:
        """LSUBGRID: OBS[nsea][8][NFRE] for ALL sea points (this rank keeps its owned rows); None switches it off.  The
        weights change: the next step re-runs the CTUW checks (and rebuilds W in the stored-weight scheme)."""
        if obs_global is None:
            self.ctx.set_obstructions(None)
        else:
            own = np.ascontiguousarray(np.asarray(obs_global)[self.dom.lo:self.dom.hi], dtype=self.npdt)
            self.ctx.set_obstructions(torch.from_numpy(own).to(self.dev))
        self.weights_ready = False

    def set_environment(self, depth_ext, u_ext, v_ext, omosnh2kd_ext, wavnum_ext) -> None:
        """DEPTH / UCUR / VCUR [nrows] and OMOSNH2KD / WAVNUM [nrows][NFRE] including halo and land rows (what PROENVHALO
        assembles): a new current field makes the next step rebuild the dot terms and re-check the weights (LLUPDTTD /
        LUPDTWGHT, propag_wam.F90:175-236)."""
        for dst, src in ((self.depth_ext, depth_ext), (self.u_ext, u_ext), (self.v_ext, v_ext), (self.omosnh2kd_ext, omosnh2kd_ext),
                         (self.wavnum_ext, wavnum_ext)):
            dst.copy_(torch.as_tensor(np.ascontiguousarray(src, dtype=self.npdt)))
        self.weights_ready = False

    # ---- PROENVHALO (proenvhalo.F90:63-107) on the device
    def proenvhalo_pack(self, depth, ucur, vcur, omosnh2kd) -> torch.Tensor:
        """Owned DEPTH / UCUR / VCUR [n] and OMOSNH2KD [n][NFRE] (device tensors or arrays) + the WAVNUM / CGROUP of the device-resident
        WVPRPT rows -> the owned rows of BUFFER_EXT [nrows][1][3 NFRE + 3]; its halo rows are to be exchanged like spectra."""
        if not self.irefra:
            raise ValueError("PROENVHALO serves the refraction set-up (IREFRA = 1, 2, 3)")
        NFRE = self.cfg.nfre
        if getattr(self, "env_buf", None) is None:
            self.env_buf = torch.zeros((self.nrows - 1, 1, 3 * NFRE + 3), dtype=self.dtype, device=self.dev)
            land = syn.depth_props(np.array([float(self.t.BATHYMAX)]), self.t, self.npdt)      # WVPRPT_LAND (initdpthflds.F90:85-88)
            row = np.concatenate([land["WAVNUM"][0], land["CGROUP"][0], land["OMOSNH2KD"][0], [float(self.t.BATHYMAX), 0.0, 0.0]]).astype(self.npdt)
            self.env_land = torch.from_numpy(row).to(self.dev)
        a = [torch.as_tensor(np.ascontiguousarray(x, dtype=self.npdt)).to(self.dev) if not torch.is_tensor(x) else x.to(self.dev, self.dtype).contiguous()
             for x in (depth, ucur, vcur, omosnh2kd)]
        assert a[0].shape[0] == self.n and tuple(a[3].shape) == (self.n, NFRE)
        self.ctx.proenvhalo_pack(self.n, self.wvprpt, a[3], a[0], a[1], a[2], self.env_buf)
        return self.env_buf

    def proenvhalo_unpack(self) -> None:
        self.ctx.proenvhalo_unpack(self.nrows - 1, self.env_buf, self.env_land, self.wavnum_ext, self.cgroup_ext, self.omosnh2kd_ext,
                                   self.depth_ext, self.u_ext, self.v_ext)
        self.weights_ready = False      # LLUPDTTD / LUPDTWGHT (propag_wam.F90:175-236)

    def proenvhalo(self, depth, ucur, vcur, omosnh2kd) -> None:
        """New depth / currents on a decomposed grid without a host round trip: pack, MPEXCHNG(BUFFER_EXT, 3*NFRE_RED+5, 1, 1), unpack."""
        self.halo(self.proenvhalo_pack(depth, ucur, vcur, omosnh2kd))
        self.proenvhalo_unpack()

    # ---- CTUWUPDT (ctuwupdt.F90:220-256): weights for the (sub-)step structure
    def build_weights(self) -> int:
        c = self.cfg
        self.cflfail.zero_()
        if self.irefra:
            # PROPDOT, then CTUWINI + CTUWDRV per frequency range (checks only: the weights are rebuilt inside PROPAGS2)
            self.ctx.propdot(self.gd, self.depth_ext, self.u_ext, self.v_ext, self.refr)
            a = (self.gd, self.cgroup_ext, self.omosnh2kd_ext, self.wavnum_ext, self.refr, self.cflfail)
            if self.ifrelfmax <= 0:
                self.ctx.ctuw_refra(*a, float(c.idelpro), 1, c.nfre_red, llcflcuroff=self.llcflcuroff, frange=0)
            else:
                self.ctx.ctuw_refra(*a, float(self.delpro_lf), 1, self.ifrelfmax, llcflcuroff=self.llcflcuroff, frange=0)
                if self.ifrelfmax < c.nfre_red:
                    self.ctx.ctuw_refra(*a, float(c.idelpro), self.ifrelfmax + 1, c.nfre_red, llcflcuroff=self.llcflcuroff, frange=1)
            self.weights_ready = True
            return int(self.cflfail.sum().item())
        # weights == "otf": w is None, CTUW only snaps WLAT/WCOR near land and runs the CFL / range checks
        if self.ifrelfmax <= 0:
            self.ctx.ctuw(self.gd, self.cgroup_ext, self.w, self.cflfail, float(c.idelpro), 1, c.nfre_red)
        else:
            self.ctx.ctuw(self.gd, self.cgroup_ext, self.w, self.cflfail, float(self.delpro_lf), 1, self.ifrelfmax)
            if self.ifrelfmax < c.nfre_red:
                self.ctx.ctuw(self.gd, self.cgroup_ext, self.w, self.cflfail, float(c.idelpro), self.ifrelfmax + 1, c.nfre_red)
        self.weights_ready = True
        return int(self.cflfail.sum().item())

    def _halo_finish_timed(self, reqs) -> None:
        """halo.finish between two events on the compute stream when the caller collects them (bench.py: what the overlap did not hide)."""
        if self.halo_events is None:
            self.halo.finish(reqs)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.halo.finish(reqs)
        e1.record()
        self.halo_events.append((e0, e1))

    # ---- PROPAG_WAM (propag_wam.F90:166-313), IPROPAGS = 2
    def propag(self) -> None:
        if not self.weights_ready:
            nfail = self.build_weights()
            if nfail:
                raise api.EcwamHipError(f"CFL criterion violated at {nfail} points (ctuwdrv.F90:128-146)")
        c, g = self.cfg, self.gd
        lf = 0 < self.ifrelfmax < c.nfre_red

        def advect_rows(k0, k1, m1, m2, delpro, copy_rest, split=0, src=None):
            if k1 <= k0:
                return
            tiled = self.tiles2d and (k0, k1) == (0, self.n)
            if tiled:                                        # the range indexes the entries of the (padded) tile order
                k1 = int(self.order.shape[0])
            if src is not None and src is not self.fl1:      # compact fast-wave rows -> fast-wave slots of FL3
                self.ctx.propags2_otf(src, self.fl3, g, self.cgroup_ext, delpro, k0, k1, m1, m2, copy_rest=False, order=self.order, tiles2d=tiled)
                return
            if self.irefra:
                self.ctx.propags2_refra(self.fl1, self.fl3, g, self.cgroup_ext, self.omosnh2kd_ext, self.wavnum_ext, self.refr, delpro, k0,
                                        k1, m1, m2, copy_rest=copy_rest, frange=int(0 < self.ifrelfmax < m1))
            elif self.weights == "stored":
                self.ctx.propags2(self.fl1, self.fl3, g["klon"], g["klat"], g["kcor"], self.w, k0, k1, m1, m2, copy_rest=copy_rest)
            else:
                self.ctx.propags2_otf(self.fl1, self.fl3, g, self.cgroup_ext, delpro, k0, k1, m1, m2, copy_rest=copy_rest,
                                      order=self.order, tiles2d=tiled, ifrelfmax=split, delpro_lf=self.delpro_lf if split else None,
                                      gout=self.g1 if split else None)   # the fast waves also into the compact buffer

        def advect(m1, m2, delpro, copy_rest, split=0, rows=None, src=None):
            k0, k1 = rows if rows is not None else (0, self.n)
            advect_rows(k0, k1, m1, m2, delpro, copy_rest, split, src)

        def exchange_and_advect(passes, src=None):
            """MPEXCHNG + PROPAGS2 (propag_wam.F90:166,247-313) with the exchange hidden behind the interior: post the halo
            exchange, advect the rows that read no halo row, wait, advect the two ends of the band.  src: the buffer whose
            halo rows are exchanged and that the stencil reads (FL1, or the compact fast-wave buffer)."""
            src = self.fl1 if src is None else src
            overlap = self.dom.nranks > 1 and self.order is None
            if not overlap:
                self._halo_finish_timed(self.halo.start(src))
                for a in passes:
                    advect(*a, src=src)
                return
            ia, ib = self.interior
            reqs = self.halo.start(src)
            for a in passes:
                advect(*a, rows=(ia, ib), src=src)
            self._halo_finish_timed(reqs)
            for a in passes:
                advect(*a, rows=(0, ia), src=src)
                advect(*a, rows=(ib, self.n), src=src)

        otf_plain = self.weights == "otf" and not self.irefra
        if lf and otf_plain and self.g1 is not None and self.fast_mode == "compact" and self.order is None:
            self._propag_fast_compact(advect_rows)
            return
        if self.weights == "stored" or self.ifrelfmax <= 0:
            exchange_and_advect([(1, c.nfre_red, float(c.idelpro), True)])      # stored W already carries the per-range time steps
        elif otf_plain:
            # fast and slow waves in ONE pass, each with its own time step (the reference calls PROPAGS2 per range)
            exchange_and_advect([(1, c.nfre_red, float(c.idelpro), True, self.ifrelfmax if lf else c.nfre_red)])
        else:
            passes = [(1, self.ifrelfmax, float(self.delpro_lf), True)]
            if lf:
                passes.append((self.ifrelfmax + 1, c.nfre_red, float(c.idelpro), False))
            exchange_and_advect(passes)
        if lf:
            nstep_lf = int(round(float(c.idelpro) / float(self.delpro_lf)))
            for isub in range(2, nstep_lf + 1):
                # FL1_EXT(:,:,1:IFRELFMAX) <- FL3_EXT ; exchange ; PROPAGS2 on the fast waves only
                if self.g1 is not None:
                    if isub > 2:      # the first pass wrote the compact buffer itself
                        self.ctx.copy_freq_range(self.fl3, self.g1, self.n, 1, self.ifrelfmax)
                    exchange_and_advect([(1, self.ifrelfmax, float(self.delpro_lf), False)], src=self.g1)
                else:
                    self.ctx.copy_freq_range(self.fl3, self.fl1, self.n, 1, self.ifrelfmax)
                    exchange_and_advect([(1, self.ifrelfmax, float(self.delpro_lf), False)])
        self.fl1, self.fl3 = self.fl3, self.fl1
        self.gfast_valid = False      # (the rows mode and the stored-weight / refraction paths overwrite or bypass the compact rows)

    def _propag_fast_compact(self, advect_rows) -> None:
        """PROPAG_WAM with fast-wave sub-steps (propag_wam.F90:247-313) in the order that writes no frequency sub-range into full rows:
        the fast waves do not depend on the slow ones, so their sub-steps 1 .. NSTEP_LF-1 run first, compact -> compact (rows 4.5 x
        shorter at IFRELFMAX = 5, for the stencil and for MPEXCHNG), and the one full pass takes their last state as the input of the last
        sub-step next to the slow waves of FL1 and writes complete FL3 rows (+ the compact copy the next advection step starts from)."""
        c, g = self.cfg, self.gd
        lfm, dlf = self.ifrelfmax, float(self.delpro_lf)
        nstep_lf = int(round(float(c.idelpro) / dlf))
        lfp = int(self.g1.shape[2])
        if not self.gfast_valid:      # (after IMPLSCH or a new initial state) FL1's first LFP frequencies -> compact rows
            self.ctx.copy_freq_range(self.fl1, self.g1, self.n, 1, lfp)
        ga, gb = self.g1, self.g2
        overlap = self.dom.nranks > 1
        ia, ib = self.interior if overlap else (0, self.n)

        def run(passes, exchanges):
            """post the exchanges, advect the rows that read no halo row, wait, advect the two ends of the band"""
            reqs = [self.halo.start(x) for x in exchanges] if self.dom.nranks > 1 else []
            passes(ia, ib)
            for r in reqs:
                self._halo_finish_timed(r)
            if overlap:
                passes(0, ia)
                passes(ib, self.n)

        for _ in range(nstep_lf - 1):
            src, dst = ga, gb
            run(lambda k0, k1: k1 > k0 and self.ctx.propags2_otf(src, dst, g, self.cgroup_ext, dlf, k0, k1, 1, lfm, copy_rest=True), [src])
            ga, gb = gb, ga
        # the full pass: slow waves from FL1 with IDELPRO, the fast waves' last sub-step from the compact rows with DELPRO_LF
        run(lambda k0, k1: k1 > k0 and self.ctx.propags2_otf(self.fl1, self.fl3, g, self.cgroup_ext, float(c.idelpro), k0, k1, 1, c.nfre_red, copy_rest=True,
                                                            ifrelfmax=lfm, delpro_lf=dlf, gin=ga, gout=gb), [ga, self.fl1])      # the short rows first (as WAMINTGR_HIP posts them)
        self.g1, self.g2 = gb, ga
        self.gfast_valid = True
        self.fl1, self.fl3 = self.fl3, self.fl1

    def newwind(self) -> None:
        if self.ff_next is not None:
            self.ctx.newwind(self.ff, self.ff_next)

    def _fast_sink(self) -> bool:
        """IMPLSCH / NOSOURCE leave the fast waves of their result in the compact rows the next advection step starts from."""
        on = self.g1 is not None and self.fast_mode == "compact" and self.order is None
        self.ctx.set_fastwave_copy(self.g1 if on else None)
        return on

    def implsch(self, wam2nemo=None) -> None:
        """(Whoever writes FL1 by another route -- a direct ctx.implsch, a tensor assignment -- resets `gfast_valid`: the compact
        fast-wave rows then no longer describe FL1 and the next propag() extracts them again.)"""
        self.gfast_valid = False
        on = self._fast_sink()
        try:
            self.ctx.implsch(0, self.n, self.fl1, self.wvprpt, self.ff, self.intf, self.mij, self.xllws, wam2nemo=wam2nemo)
        finally:
            if on:
                self.ctx.set_fastwave_copy(None)      # the library keeps no pointer into a buffer this object may swap or free
        self.gfast_valid = on

    def nosource(self) -> None:
        """NO SOURCE TERM CONTRIBUTION (wamintgr.F90:152-160, LLSOURCE = F)."""
        self.gfast_valid = False
        on = self._fast_sink()
        try:
            self.ctx.nosource(0, self.n, self.fl1, self.mij, self.xllws)
        finally:
            if on:
                self.ctx.set_fastwave_copy(None)
        self.gfast_valid = on

    # ---- the 1:1 step as one kernel (round 6): PROPAGS2 inside IMPLSCH's tile load (ecwam_hip_propags2_implsch)
    def fused_available(self) -> bool:
        """The one-kernel step covers this model: a build exists (36 x 36, common IMPLSCH builds) and the advection is IREFRA = 0 with
        on-the-fly weights in the natural row order, with or without sub-grid obstructions; fast-wave sub-steps run on compact rows (the
        last one inside the kernel)."""
        lf = 0 < self.ifrelfmax < self.cfg.nfre_red
        return (self.ctx.fused_supported(fast_waves=lf, obstructions=self.ctx.has_obstructions) and not self.irefra and self.weights == "otf"
                and self.order is None and (not lf or (self.g1 is not None and self.fast_mode == "compact")))

    def step_fused(self, wam2nemo=None, flags: int = 0) -> None:
        """PROPAG_WAM + NEWWIND + IMPLSCH of one step (wamintgr.F90:94-146) with the advection done by the source-term kernel's tile load.
        NEWWIND touches the forcing only and runs first; the halo exchange is posted, the rows that read no halo row are integrated while
        it runs, then the two ends of the band (propag_wam.F90:166, mpexchng.F90)."""
        if not self.weights_ready:
            nfail = self.build_weights()
            if nfail:
                raise api.EcwamHipError(f"CFL criterion violated at {nfail} points (ctuwdrv.F90:128-146)")
        self.newwind()
        c = self.cfg
        lf = 0 < self.ifrelfmax < c.nfre_red
        multi = self.dom.nranks > 1
        ia, ib = self.interior if multi else (0, self.n)
        ga = gb = None
        kw = {}
        if lf:
            # the fast waves' sub-steps 1 .. NSTEP_LF-1 on compact rows (propag_wam.F90:285-313, as _propag_fast_compact); their last sub-step
            # and the slow waves' step are the tile load of the source-term kernel, which also leaves the new fast waves in compact rows
            lfm, dlf = self.ifrelfmax, float(self.delpro_lf)
            nstep_lf = int(round(float(c.idelpro) / dlf))
            lfp = int(self.g1.shape[2])
            if not self.gfast_valid:
                self.ctx.copy_freq_range(self.fl1, self.g1, self.n, 1, lfp)
            ga, gb = self.g1, self.g2
            for _ in range(nstep_lf - 1):
                reqs = self.halo.start(ga) if multi else []
                self.ctx.propags2_otf(ga, gb, self.gd, self.cgroup_ext, dlf, ia, ib, 1, lfm, copy_rest=True)
                if multi:
                    self._halo_finish_timed(reqs)
                    for k0, k1 in ((0, ia), (ib, self.n)):
                        if k1 > k0:
                            self.ctx.propags2_otf(ga, gb, self.gd, self.cgroup_ext, dlf, k0, k1, 1, lfm, copy_rest=True)
                ga, gb = gb, ga
            kw = dict(ifrelfmax=lfm, delpro_lf=dlf, gin=ga)
            self.ctx.set_fastwave_copy(gb)

        def rows(k0, k1):
            if k1 > k0:
                self.ctx.propags2_implsch(self.fl1, self.fl3, self.gd, self.cgroup_ext, float(c.idelpro), k0, k1, self.wvprpt, self.ff, self.intf,
                                          self.mij, self.xllws, 1, c.nfre_red, wam2nemo=wam2nemo, flags=flags, **kw)

        try:
            if multi:
                reqs = ([self.halo.start(ga)] if lf else []) + [self.halo.start(self.fl1)]      # the short rows first
                rows(ia, ib)
                for r in reqs:
                    self._halo_finish_timed(r)
                rows(0, ia)
                rows(ib, self.n)
            else:
                rows(0, self.n)
        finally:
            if lf:
                self.ctx.set_fastwave_copy(None)
        self.fl1, self.fl3 = self.fl3, self.fl1
        if lf:
            self.g1, self.g2 = gb, ga
        self.gfast_valid = lf

    def step(self, advect: bool = True, source: bool = True, llsource: bool = True, fused: bool = False) -> None:
        if fused and advect and source and llsource and self.fused_available():
            self.step_fused()
            return
        if advect:
            self.propag()
        self.newwind()
        if source:
            if llsource:
                self.implsch()
            else:
                self.nosource()

    # ---- restart spectra in the reference's file layout (writefl.F90:110-118, one record per rank)
    def write_restart(self, path: str) -> None:
        from . import restart
        restart.write_fl(path, self.fl1[: self.n].cpu().numpy())

    def read_restart(self, path: str) -> None:
        from . import restart
        a = restart.read_fl(path, self.n, self.cfg.nang, self.cfg.nfre, self.npdt)
        self.fl1[: self.n] = torch.from_numpy(a).to(self.dev)
        self.gfast_valid = False

    # ---- OUTBS subset on the device: [n][5] = swh, mean direction, mean period, EM, peak period; norms = OUTWNORM (avg, min, max, count)
    def outbs(self) -> torch.Tensor:
        out = torch.zeros((self.n, 5), dtype=self.dtype, device=self.dev)
        self.ctx.outbs(0, self.n, self.fl1, out)
        return out

    def swh_norm(self):
        return self.ctx.outwnorm(self.outbs(), 0, self.n)

    # ---- diagnostics used by tests/bench (swh = 4 sqrt(EM), outbs.F90 / semean.F90)
    def swh(self) -> torch.Tensor:
        dfim = torch.from_numpy(np.asarray(self.t.DFIM, dtype=np.float64)).to(self.dev)
        e = (self.fl1[: self.n].double().sum(1) * dfim).sum(1)
        return 4.0 * torch.sqrt(e)
