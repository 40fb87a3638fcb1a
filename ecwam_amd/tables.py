"""Host-side spectral tables and run parameters for the WAMINTGR hot path.

These are the module-level globals the reference kernels read:
YOWFRED, YOWINDN, YOWPHYS, YOWPCONS, YOWCOUP, YOWICE, YOWTABL.  In a drop-in build the Fortran
host hands them over through ``ecwam_hip_set_tables`` (include/ecwam_hip.h); for the standalone
harness (tests, bench) this module computes them, following

  mfredir.F90:90-129 + mfr.F90:42-48   FR, DFIM, TH, COSTH, SINTH, DELTH
  initmdl.F90:436-508                  DFIMOFR, DFIMFR, ZPIFR, FR5, COFRM4, FLMAX, RHOWG_DFIM, DFIM_SIM, NFRE_ODD
  setwavphys.F90:115-202               IPHYS=1 constants by (LLGCBZ0, LLNORMAGAM)
  init_x0tauhf.F90:65-100              BETAMAXOXKAPPA2, X0TAUHF, WTAUHF
  tabu_swellft.F90:64-83 (+kerkei/kzeone: Kelvin functions ker/kei, here from scipy.special)
  nlweigt.F90:94-262, inisnonlin.F90:89-270, jafu.F90   DIA index/weight tables
  init_sdiss_ardh.F90:67-96            INDICESSAT, SATWEIGHTS
  initgc.F90:65-109                    gravity-capillary tables
  ctuwupdt.F90:97-161                  KPM, JXO, JYO, KCR upwind selectors

The reference evaluates every table in its working precision JWRB (sp or dp build,
parkind_wave.F90:23-35); the recursion FR(M)=FRATIO*FR(M-1) alone makes sp tables differ from dp
ones by ~1e-6 and the DIA interpolation weights (differences of nearly equal numbers) by ~1e-4.
To be a faithful stand-in for the Fortran host, the arithmetic below is therefore carried out in
the requested dtype with the reference's operation order (numpy scalars round after each op).
"""
from __future__ import annotations

import dataclasses

import os

import numpy as np

JTOT_TAUHF = 19  # yowcoup.F90:60
IAB = 200        # yowtabl.F90:25


@dataclasses.dataclass
class Config:
    """Namelist/flag subset read on the hot path (ecwam_run_model.sh:211-272, mpuserin.F90:548-808)."""

    nang: int = 36
    nfre: int = 36
    nfre_red: int = 36
    ifre1: int = 3
    fr1: float = 4.177248e-02
    idelt: int = 900       # source-term time step [s]
    idelpro: int = 900     # advection time step [s]
    ximp: float = 1.0
    iphys: int = 1
    isnonlin: int = 0
    irefra: int = 0
    icode: int = 3
    llgcbz0: bool = False
    llnormagam: bool = False
    llcapchnk: bool = True
    lbiwbk: bool = True
    licerun: bool = True
    lmaskice: bool = True
    lwamrsetci: bool = True
    lciwa1: bool = False
    lciwa2: bool = False
    lciwa3: bool = False
    lciscal: bool = False
    lwvflx_snl: bool = True
    lwflux: bool = False
    lwfluxout: bool = True
    lwnemocou: bool = False
    lwcou: bool = False
    lwcouast: bool = True
    lwnemocouwrs: bool = False
    lwnemocouibr: bool = False
    lwnemotauoc: bool = False
    lwnemocousend: bool = True     # mpuserin.F90:721-723
    lwnemocoustk: bool = True
    lwnemocoustrn: bool = False
    zalpfacb: float = 1.0          # mpuserin.F90:780-786 (namelist NALINE)
    zalpfacx: float = 1.0
    zalpwrs: float = 1.0
    zibrw_thrsh: float = 0.5
    wspmin: float = -1.0
    rnu: float = 1.5e-5            # runwam.F90:232
    rnum: float = 0.11 * 1.5e-5    # runwam.F90:233

    def validate(self) -> None:
        if self.iphys not in (0, 1):
            raise NotImplementedError("IPHYS must be 0 (Janssen) or 1 (Ardhuin)")
        if self.isnonlin not in (0, 1, 2):
            raise ValueError("ISNONLIN must be 0, 1 (TRANSF) or 2 (TRANSF_SNL with PEAK_ANG)")
        if self.irefra not in (0, 1, 2, 3):
            raise ValueError("IREFRA must be 0 (none), 1 (depth), 2 (currents) or 3 (depth + currents)")
        if self.icode not in (1, 2, 3):
            raise ValueError("ICODE must be 3 (10 m wind), 1 or 2 (friction velocity / stress forcing)")
        if self.nfre_red <= 0:
            self.nfre_red = self.nfre
        if not (8 <= self.nfre <= 48 and 4 <= self.nang <= 48 and self.nfre_red <= self.nfre):
            raise ValueError("spectral dimensions out of the supported range")


CIDEAC_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cideac_kohout_meylan.txt")


def powi(x, n: int):
    """Fortran X**N for integer N (binary powering, the order compilers expand it in)."""
    m = abs(n)
    y = x if (m % 2) else x * 0 + 1
    while m > 1:
        m >>= 1
        x = x * x
        if m % 2:
            y = y * x
    return 1 / y if n < 0 else y


class Tables:
    """All read-only tables in the working precision ``dtype`` (np.float32 or np.float64)."""

    def __init__(self, cfg: Config, dtype=np.float64):
        cfg.validate()
        self.cfg = cfg
        self.dtype = T = np.dtype(dtype).type
        c = cfg
        NANG, NFRE = c.nang, c.nfre
        # ---- YOWPCONS (yowpcons.F90:19-66, iniwcst.F90:54-69)
        self.G = T(9.806)
        self.GM1 = T(0.101978381)
        self.PI = T(4.0) * np.arctan(T(1.0))
        self.ZPI = T(2.0) * self.PI
        self.ZPI4GM1 = powi(self.ZPI, 4) / self.G
        self.ZPI4GM2 = powi(self.ZPI, 4) / (self.G * self.G)
        self.RAD = self.PI / T(180.0)
        self.DEG = T(180.0) / self.PI
        self.CIRC = T(40007993.95)
        self.R = self.CIRC / self.ZPI * T(1.0)
        self.EPSMIN = T(0.1e-32)
        self.ROWATER = T(1000.0)
        self.ROWATERM1 = T(1.0) / self.ROWATER
        self.ROAIR = T(1.225)
        self.GAM_SURF = T(0.0717)
        self.SURFT = self.GAM_SURF / self.ROWATER
        self.EPSUS = T(1.0e-6)
        self.EPSU10 = np.sqrt(T(1.0e-3))
        self.ACD, self.BCD = T(8.0e-4), T(8.0e-5)
        self.ACDLIN, self.BCDLIN = T(0.0008), T(0.00047)
        self.CDMAX = T(0.0025)
        self.TAUOCMIN, self.TAUOCMAX = T(0.01), T(50.0)
        self.PHIEPSMIN, self.PHIEPSMAX = T(-3276.80), T(-0.05)
        self.WSEMEAN_MIN = T(0.001)
        # ---- YOWFRED (yowfred.F90:50-82)
        self.FRATIO = T(1.1)
        self.WETAIL, self.FRTAIL, self.WP1TAIL = T(0.25), T(0.2), T(1.0) / T(3.0)
        self.COEF4 = T(5.0e-07)
        self.FRIC = T(28.0)
        FR = np.zeros(NFRE, dtype=T)
        FR[c.ifre1 - 1] = T(c.fr1)
        for m in range(c.ifre1 - 1, 0, -1):
            FR[m - 1] = FR[m] / self.FRATIO
        for m in range(c.ifre1 + 1, NFRE + 1):
            FR[m - 1] = self.FRATIO * FR[m - 2]
        self.FR = FR
        self.DELTH = self.ZPI / T(NANG)
        self.TH = np.arange(NANG).astype(T) * self.DELTH + T(0.5) * self.DELTH
        self.COSTH, self.SINTH = np.cos(self.TH), np.sin(self.TH)
        co1 = T(0.5) * (self.FRATIO - T(1.0)) * self.DELTH
        self.DFIM = np.empty(NFRE, dtype=T)
        self.DFIM[0] = co1 * FR[0]
        self.DFIM[1:-1] = co1 * (FR[1:-1] + FR[:-2])
        self.DFIM[-1] = co1 * FR[-2]
        self.DFIMOFR = self.DFIM / FR
        self.DFIMFR = self.DFIM * FR
        self.ZPIFR = self.ZPI * FR
        self.FR5 = powi(FR, 5)
        self.COFRM4 = self.COEF4 * self.G / powi(FR, 4)
        self.FLOGSPRDM1 = T(1.0) / np.log10(self.FRATIO)
        xl = np.log(self.FRATIO)
        self.RHOWG_DFIM = self.ROWATER * self.G * self.DELTH * xl * FR
        self.RHOWG_DFIM[0] = T(0.5) * self.ROWATER * self.G * self.DELTH * xl * FR[0]
        self.RHOWG_DFIM[-1] = T(0.5) * self.ROWATER * self.G * self.DELTH * xl * FR[-1]
        self.NFRE_ODD = NFRE - 1 + (NFRE % 2)
        sim = np.zeros(NFRE, dtype=T)
        sim[0] = self.DELTH * xl * FR[0] / T(3.0)
        for mm in range(2, self.NFRE_ODD, 2):  # M=2,NFRE_ODD-1,2 (1-based)
            sim[mm - 1] = T(4.0) * self.DELTH * xl * FR[mm - 1] / T(3.0)
            sim[mm] = T(2.0) * self.DELTH * xl * FR[mm] / T(3.0)
        sim[self.NFRE_ODD - 1] = self.DELTH * xl * FR[self.NFRE_ODD - 1] / T(3.0)
        self.DFIM_SIM = sim
        # ---- YOWPHYS constants + setwavphys
        self.XKAPPA, self.XNLEV, self.ALPHAMAX = T(0.40), T(10.0), T(0.11)
        self.SWELLF, self.SWELLF2, self.SWELLF3 = T(0.66), T(-0.018), T(0.022)
        self.SWELLF5, self.SWELLF6 = T(1.2), T(1.0)
        self.ABMIN, self.ABMAX = T(0.3), T(8.0)
        self.SDSBR, self.ISDSDTH, self.ISB, self.IPSAT = T(9.0e-4), 80, 2, 2
        self.SSDSC2, self.SSDSC3, self.SSDSC4 = T(-2.2e-5), T(0.0), T(1.0)
        self.SSDSC6, self.MICHE = T(0.3), T(1.0)
        self.RNU, self.RNUM = T(c.rnu), T(c.rnum)
        self.ZALP, self.TAILFACTOR, self.TAILFACTOR_PM = T(0.008), T(2.5), T(3.0)
        if NANG <= 24:
            self.ANG_GC_A, self.ANG_GC_B, self.ANG_GC_C = T(0.40), T(0.60), T(3.0)
        else:
            self.ANG_GC_A, self.ANG_GC_B, self.ANG_GC_C = T(0.35), T(0.65), T(3.0)
        self.RN1_RN = T(0.25)
        if c.llgcbz0:
            self.ALPHA, self.ALPHAMIN, self.CHNKMIN_U, self.ALPHAPMAX = T(0.0055), T(0.0001), T(28.0), T(0.03)
            self.DELTA_THETA_RN, self.DTHRN_A, self.DTHRN_U = T(0.75), T(0.60), T(33.0)
            self.Z0TUBMAX, self.Z0RAT, self.SWELLF4, self.SWELLF7 = T(0.05), T(0.02), T(1.15e05), T(4.32e05)
            self.SSDSC5 = T(0.0)
            self.BETAMAX, self.TAUWSHELTER = (T(1.39), T(0.0)) if c.llnormagam else (T(1.44), T(0.25))
        else:
            self.ALPHA, self.ALPHAPMAX = T(0.0065), T(0.031)
            self.DELTA_THETA_RN, self.DTHRN_A, self.DTHRN_U = T(0.75), T(0.60), T(200.0)
            self.Z0TUBMAX, self.Z0RAT, self.SWELLF4, self.SWELLF7 = T(0.0005), T(0.04), T(1.5e05), T(3.6e05)
            self.SSDSC5 = T(0.0)
            if c.llnormagam:
                self.BETAMAX, self.TAUWSHELTER, self.ALPHAMIN, self.CHNKMIN_U = T(1.39), T(0.0), T(0.0005), T(30.0)
            else:
                self.BETAMAX, self.TAUWSHELTER, self.ALPHAMIN, self.CHNKMIN_U = T(1.40), T(0.25), T(0.0001), T(33.0)
        self.SWELLF7M1 = T(1.0) / self.SWELLF7
        self.EGRCRV, self.AFCRV, self.BFCRV = T(1065.0), T(2.453e-4), T(-3.1236)
        self.IDAMPING = 1                                                      # mpuserin.F90:609
        self.CDIS, self.DELTA_SDIS, self.CDISVIS = T(0.0), T(0.0), T(0.0)
        if c.iphys == 0:   # Janssen wind input + WAM cycle 4 dissipation, setwavphys.F90:46-112
            self.ZALP, self.TAILFACTOR, self.ALPHAMIN, self.ALPHAPMAX, self.TAUWSHELTER = T(0.008), T(2.5), T(0.0001), T(0.03), T(0.0)
            self.DELTA_THETA_RN, self.DTHRN_A, self.DTHRN_U, self.RN1_RN, self.TAILFACTOR_PM = T(0.75), T(0.80), T(33.0), T(0.25), T(0.0)
            if c.llgcbz0:
                self.ALPHA, self.CHNKMIN_U = T(0.0055), T(28.0)
                self.BETAMAX = T(1.32) if c.llnormagam else T(1.25)
                self.CDIS, self.DELTA_SDIS, self.CDISVIS = T(-1.3), T(0.6), T(-4.0)
            else:
                self.ALPHA, self.CHNKMIN_U, self.BETAMAX = T(0.0065), T(33.0), T(1.20)
                self.CDIS, self.DELTA_SDIS, self.CDISVIS = T(-1.33), T(0.5), T(0.0)
            self.EGRCRV, self.AFCRV, self.BFCRV = T(1108.0), T(4.0e-4), T(-3.0)
        self.FLMAX = (self.ALPHAPMAX / self.PI) / (self.ZPI4GM2 * self.FR5)
        # ---- init_x0tauhf
        self.BETAMAXOXKAPPA2 = self.BETAMAX / (self.XKAPPA * self.XKAPPA)
        self.BMAXOKAP = self.DELTA_THETA_RN * self.BETAMAXOXKAPPA2 / self.XKAPPA
        self.GAMNCONST = self.BMAXOKAP * T(0.5) * powi(self.ZPI, 4) * self.GM1 * self.GM1 * self.GM1
        alph = self.ALPHAMIN if (c.llgcbz0 or c.llcapchnk or c.llnormagam) else self.ALPHA
        x0 = T(0.005)
        for _ in range(30):
            ff = np.exp(self.XKAPPA / (x0 + self.ZALP))
            f = alph * x0 * x0 * ff - T(1.0)
            if f == T(0.0):
                break
            q = x0 / (x0 + self.ZALP)
            df = alph * ff * (T(2.0) * x0 - self.XKAPPA * q * q)
            x0 = x0 - f / df
        self.X0TAUHF = x0
        const1 = self.BETAMAXOXKAPPA2 / T(3.0)
        w = np.full(JTOT_TAUHF, const1, dtype=T)
        w[1:-1:2] = T(4.0) * const1
        w[2:-1:2] = T(2.0) * const1
        self.WTAUHF = w
        self.EPS1 = T(0.00001)
        # ---- thresholds (userin.F90:913-918, 957-976)
        self.WSPMIN = T(c.wspmin) if c.wspmin > 0 else (T(0.3) if c.llgcbz0 else T(1.0))
        self.FLMIN = T(0.00001)
        if c.lmaskice:
            self.CITHRSH, self.CIBLOCK, self.CITHRSH_TAIL, self.CDICWA = T(0.3), T(0.0), T(0.3), T(0.0)
        else:
            self.CITHRSH, self.CIBLOCK, self.CITHRSH_TAIL = T(1.0), T(1.0), T(0.1)
            self.CDICWA = T(0.01) if c.lciwa2 else T(0.0)          # ice-water drag coefficient, userin.F90:971-977
        self.ZALPWRS, self.ZALPFACB, self.ZALPFACX, self.ZIBRW_THRSH = T(c.zalpwrs), T(c.zalpfacb), T(c.zalpfacx), T(c.zibrw_thrsh)
        self._cigetdeac()
        self.GAM_B_J, self.BATHYMAX, self.WSPMIN_RESET_TAUW = T(0.8), T(998.999), T(4.0)
        self._swellft()
        self._nlweigt()
        self._sdiss()
        self._gc()
        self._ctu_selectors()

    # tabu_swellft.F90: friction factor table over log10(a/z0); Kelvin functions ker/kei
    def _cigetdeac(self) -> None:
        """cigetdeac.F90:64-82,553-559: SDICE1's table of ln(attenuation per floe), CIDEAC(IT,IH), period TICMIN+(IT-1)*DTIC,
        thickness HICMIN+(IH-1)*DHIC.  Periods 6..16 s are tabulated data (ecwam_amd/data/), period 1 s is an assumed linear
        ramp -2 .. -1 over the thickness range, periods 2..5 s are interpolated between the two.  Stored [IH][IT]."""
        T = self.dtype.type if hasattr(self.dtype, "type") else self.dtype
        self.NICH, self.NICT = 36, 16
        self.DHIC, self.TICMIN, self.DTIC, self.HICMIN = T(0.1), T(1.0), T(1.0), T(0.2)
        raw = np.loadtxt(CIDEAC_DATA)
        assert raw.shape == (self.NICH, 11)
        c = np.zeros((self.NICT, self.NICH), dtype=T)
        c[5:16, :] = raw.T.astype(T)
        c[0, 0], c[0, self.NICH - 1] = T(-2.0), T(-1.0)
        dhi = c[0, self.NICH - 1] - c[0, 0]
        for ih in range(2, self.NICH):
            c[0, ih - 1] = c[0, 0] + T(ih - 1) * dhi / T(self.NICH - 1)
        for ih in range(self.NICH):
            dci = c[5, ih] - c[0, ih]
            for it in range(2, 6):
                c[it - 1, ih] = c[0, ih] + dci * T(it - 1) * self.DTIC / (T(5) * self.DTIC)
        self.CIDEAC = np.ascontiguousarray(c.T)          # [IH][IT] = Fortran storage order of CIDEAC(IT,IH)

    def _swellft(self) -> None:
        from scipy.special import kei, ker

        T = self.dtype
        kappa = T(0.40)
        delab = (self.ABMAX - self.ABMIN) / T(IAB)
        l10 = np.log(T(10.0))
        out = np.empty(IAB, dtype=T)
        dzeta0 = T(0.0)
        for i in range(1, IAB + 1):
            abrlog = self.ABMIN + T(i) * delab
            abr = np.exp(abrlog * l10)
            fact = T(1.0) / abr / (T(21.2) * kappa)
            fsubw = T(0.05)
            for _ in range(100):
                fm, dm = fsubw, dzeta0
                dzeta0 = fact * np.power(fsubw, T(-0.5))
                x = float(T(2.0) * np.sqrt(dzeta0))
                kr, ki = T(ker(x)), T(kei(x))
                fsubw = T(0.08) / (kr * kr + ki * ki)
                fsubw = T(0.5) * (fm + fsubw)
                dzeta0 = T(0.5) * (dm + dzeta0)
            out[i - 1] = fsubw
        self.SWELLFT = out  # SWELLFT(1:IAB)

    @staticmethod
    def _jafu(cl, j: int, ian: int) -> int:
        ja = j + int(cl)  # truncation toward zero
        if ja <= 0:
            ja = ian + ja - 1
        if ja >= ian:
            ja = ja - ian + 1
        return ja

    def _nlweigt(self) -> None:
        T = self.dtype
        NANG, NFRE = self.cfg.nang, self.cfg.nfre
        alamd, con = T(0.25), T(3000.0)
        one = T(1.0)
        f1p1 = np.log10(self.FRATIO)
        isp = int(np.log10(one + alamd) / f1p1 + T(0.000001))
        ism = int(np.floor(np.log10(one - alamd) / f1p1 + T(0.0000001)))
        self.MFRSTLW, self.MLSTHG, self.KFRH = 1 + ism, NFRE - ism, -ism + isp + 2
        xf = powi((one + alamd) / (one - alamd), 4)
        costh3 = (one + T(2.0) * alamd + T(2.0) * alamd * alamd * alamd) / ((one + alamd) * (one + alamd))
        delphi1 = -T(180.0) / self.PI * np.arccos(costh3)
        costh4 = np.sqrt(one - xf + xf * costh3 * costh3)
        delphi2 = T(180.0) / self.PI * np.arccos(costh4)
        deltha = self.DELTH * self.DEG
        cl1, cl2 = delphi1 / deltha, delphi2 / deltha
        klp1 = NANG + 1
        ja1 = np.zeros((NANG + 2, 3), dtype=np.int64)
        ja2 = np.zeros((NANG + 2, 3), dtype=np.int64)
        ic = 1
        for kh in (1, 2):
            klh = NANG if kh == 1 else klp1
            for k in range(1, klh + 1):
                ks = k if kh == 1 else klp1 - k + 1
                if ks > NANG:
                    continue
                ja1[ks, kh] = self._jafu(T(ic) * cl1, k, klp1)
                ja2[ks, kh] = self._jafu(T(ic) * cl2, k, klp1)
            ic = -1
        cl1 = cl1 - T(int(cl1))
        cl2 = cl2 - T(int(cl2))
        acl1, acl2 = abs(cl1), abs(cl2)
        cl11, cl21 = one - acl1, one - acl2
        self.DAL1, self.DAL2 = one / powi(one + alamd, 4), one / powi(one - alamd, 4)
        K1W = np.zeros((NANG, 2), dtype=np.int32)
        K2W, K11W, K21W = K1W.copy(), K1W.copy(), K1W.copy()
        isg = 1
        for kh in (1, 2):
            cl1h, cl2h = T(isg) * cl1, T(isg) * cl2
            for k in range(1, NANG + 1):
                ks = k if kh == 1 else NANG - k + 2
                if k == 1:
                    ks = 1
                k1 = int(ja1[k, kh])
                k11 = (k1 - 1 or NANG) if cl1h < 0 else (1 if k1 + 1 > NANG else k1 + 1)
                k2 = int(ja2[k, kh])
                k21 = (k2 - 1 or NANG) if cl2h < 0 else (1 if k2 + 1 > NANG else k2 + 1)
                K1W[ks - 1, kh - 1], K11W[ks - 1, kh - 1] = k1, k11
                K2W[ks - 1, kh - 1], K21W[ks - 1, kh - 1] = k2, k21
            isg = -1
        self.K1W, self.K2W, self.K11W, self.K21W = K1W, K2W, K11W, K21W  # 1-based values
        lo, hi = self.MFRSTLW, NFRE + self.KFRH
        frlon = {mm: self.FR[mm - 1] for mm in range(1, NFRE + 1)}
        for mm in range(0, lo - 1, -1):
            frlon[mm] = frlon[mm + 1] / self.FRATIO
        for mm in range(NFRE + 1, hi + 1):
            frlon[mm] = self.FRATIO * frlon[mm - 1]
        ikp, ikp1, ikm, ikm1, fklap, fklap1, fklam, fklam1, af11 = ({} for _ in range(9))
        for mm in range(self.MFRSTLW, self.MLSTHG + 1):
            frg = frlon[mm]
            af11[mm] = con * powi(frg, 11)
            flp, flm = frg * (one + alamd), frg * (one - alamd)
            ikn = mm + isp
            ikp[mm], ikp1[mm] = ikn, ikn + 1
            fkp = frlon[ikn]
            fklap[mm] = (flp - fkp) / (frlon[ikn + 1] - fkp)
            fklap1[mm] = one - fklap[mm]
            ikn = mm + ism
            if ikn >= self.MFRSTLW:
                ikm[mm], ikm1[mm] = ikn, ikn + 1
                fkm = frlon[ikn]
                fklam[mm] = (flm - fkm) / (frlon[ikn + 1] - fkm)
                fklam1[mm] = one - fklam[mm]
            elif ikn + 1 == self.MFRSTLW:
                ikm[mm], ikm1[mm] = 1, self.MFRSTLW
                fkm = frlon[self.MFRSTLW] / self.FRATIO
                fklam[mm] = (flm - fkm) / (frlon[self.MFRSTLW] - fkm)
                fklam1[mm] = T(0.0)
            else:
                ikm[mm], ikm1[mm], fklam[mm], fklam1[mm] = 1, 1, T(0.0), T(0.0)
        frh = {i: powi(frlon[NFRE] / frlon[NFRE + i - 1], 5) for i in range(1, self.KFRH + 1)}

        # inisnonlin.F90:94-268
        def epmma(x):
            return np.exp(-min(T(1.25) * powi(x, 4), T(50.0))) * powi(x, 5)

        ftrf, frr = {}, one
        al = one / epmma(one)
        for mc in range(1, self.MFRSTLW - 1, -1):
            ftrf[mc] = al * epmma(frr)
            frr = frr * self.FRATIO
        ML = self.MLSTHG
        self.IKP = np.array([ikp[i] for i in range(1, ML + 1)], dtype=np.int32)
        self.IKP1 = np.array([ikp1[i] for i in range(1, ML + 1)], dtype=np.int32)
        self.IKM = np.array([ikm[i] for i in range(1, ML + 1)], dtype=np.int32)
        self.IKM1 = np.array([ikm1[i] for i in range(1, ML + 1)], dtype=np.int32)
        self.AF11 = np.array([af11[i] for i in range(1, ML + 1)], dtype=T)
        INL = np.zeros((ML, 5), dtype=np.int32)
        RNL = np.zeros((ML, 25), dtype=T)
        for mc in range(1, ML + 1):
            mp, mp1, mm_, mm1 = ikp[mc], ikp1[mc], ikm[mc], ikm1[mc]
            ffacp = ffacp1 = ffacm = ffacm1 = ftail = one
            ic_, ip, ip1, im, im1 = max(mc, 1), mp, mp1, mm_, mm1
            if ip < 1:
                ffacp, ip = ftrf[ip], 1
            if ip1 < 1:
                ffacp1, ip1 = ftrf[ip1], 1
            if im < self.MFRSTLW:
                ffacm, im = T(0.0), 1
            elif im < 1:
                ffacm, im = ftrf[im], 1
            if im1 < self.MFRSTLW:
                ffacm1, im1 = T(0.0), 1
            elif im1 < 1:
                ffacm1, im1 = ftrf[im1], 1
            if ip1 > NFRE:
                ffacp1 = frh[min(ip1 - NFRE + 1, self.KFRH)]
                ip1 = NFRE
                if ip > NFRE:
                    ffacp, ip = frh[ip - NFRE + 1], NFRE
                    if ic_ > NFRE:
                        ftail, ic_ = frh[ic_ - NFRE + 1], NFRE
                        if im1 > NFRE:
                            ffacm1, im1 = frh[im1 - NFRE + 1], NFRE
            INL[mc - 1] = (ic_, ip, ip1, im, im1)
            fp, fp1 = fklap[mc], fklap1[mc]
            gw2 = fp1 * ffacp * self.DAL1
            gw1, gw2 = gw2 * cl11, gw2 * acl1
            gw4 = fp * ffacp1 * self.DAL1
            gw3, gw4 = gw4 * cl11, gw4 * acl1
            fpa, fpb, fp2, fp1c = fp * cl11, fp * acl1, fp1 * acl1, fp1 * cl11
            fm, fm1 = fklam[mc], fklam1[mc]
            gw6 = fm1 * ffacm * self.DAL2
            gw5, gw6 = gw6 * cl21, gw6 * acl2
            gw8 = fm * ffacm1 * self.DAL2
            gw7, gw8 = gw8 * cl21, gw8 * acl2
            fma, fmb, fm2, fm1c = fm * cl21, fm * acl2, fm1 * acl2, fm1 * cl21
            RNL[mc - 1] = (ftail, gw1, gw2, gw3, gw4, fpa, fpb, fp2, fp1c, fpa * fpa, fpb * fpb, fp1c * fp1c, fp2 * fp2,
                           gw5, gw6, gw7, gw8, fma, fmb, fm2, fm1c, fma * fma, fmb * fmb, fm1c * fm1c, fm2 * fm2)
        self.INLCOEF, self.RNLCOEF = INL, RNL

    def _sdiss(self) -> None:
        T = self.dtype
        NANG = self.cfg.nang
        nangd = NANG // 2
        n = min(int(np.floor(T(self.ISDSDTH) * self.RAD / self.DELTH + T(0.5))), nangd - 1)  # NINT
        self.NSDSNTH = n
        dtr = (self.TH[0] + T(self.ISDSDTH) * self.RAD) - (self.TH[n] - T(0.5) * self.DELTH)
        dtr = max(T(0.0), min(dtr, self.DELTH))
        idx = np.zeros((NANG, 2 * n + 1), dtype=np.int32)
        wts = np.zeros((NANG, 2 * n + 1), dtype=T)
        for k in range(NANG):
            for j, i_int in enumerate(range(k - n, k + n + 1)):
                jj = i_int % NANG
                idx[k, j] = jj
                dl = dtr if (j == 0 or j == 2 * n) else self.DELTH
                cs = np.cos(self.TH[k] - self.TH[jj])
                wts[k, j] = dl * cs * cs  # **ISB, ISB = 2
        self.INDICESSAT, self.SATWEIGHTS = idx, wts  # 0-based indices

    def _gc(self) -> None:
        T = self.dtype
        kr, xks, xkl = T(1.2), T(0.006), T(20000.0)
        self.XLOGKRATIOM1_GC = T(1.0) / np.log(kr)
        self.SQRTGOSURFT = np.sqrt(self.G / self.SURFT)
        n = int(np.floor(np.log(xkl / xks) / np.log(kr) + T(0.5)))
        self.NWAV_GC = n
        xk = np.array([xks * powi(kr, i) for i in range(n)], dtype=T)
        om = np.sqrt(self.G * xk + self.SURFT * (xk * xk * xk))
        vg = T(0.5) / om * (self.G + T(3.0) * self.SURFT * (xk * xk))
        cc = om / xk
        xkm = T(1.0) / xk
        self.XK_GC, self.XKM_GC, self.OMEGA_GC = xk, xkm, om
        self.OMXKM3_GC = om * xkm * xkm * xkm
        self.CM_GC = T(1.0) / cc
        self.C2OSQRTVG_GC = cc * cc / np.sqrt(vg)
        self.XKMSQRTVGOC2_GC = xkm / self.C2OSQRTVG_GC
        self.OM3GMKM_GC = om * om * om / (self.G * xk)
        d = np.empty(n, dtype=T)
        dns = np.empty(n, dtype=T)
        d[0] = T(0.5) * (xk[1] - xk[0]) / self.C2OSQRTVG_GC[0]
        dns[0] = d[0]
        d[1:-1] = T(0.5) * (xk[2:] - xk[:-2]) / self.C2OSQRTVG_GC[1:-1]
        dns[1:-1] = T(0.5) * (xk[2:] - xk[1:-1]) / self.C2OSQRTVG_GC[1:-1]
        d[-1] = T(0.5) * (xk[-1] - xk[-2]) / self.C2OSQRTVG_GC[-1]
        dns[-1] = d[-1]
        self.DELKCC_GC_NS = dns
        self.DELKCC_OMXKM3_GC = d * self.OMXKM3_GC

    def _ctu_selectors(self) -> None:
        NANG = self.cfg.nang
        k = np.arange(NANG)
        self.KPM = np.stack([(k - 1) % NANG, k, (k + 1) % NANG], axis=1).astype(np.int32)  # 0-based
        north = self.COSTH >= 0  # northward propagation: the upwind latitude neighbour is the southern row (index 1)
        east = self.SINTH >= 0
        # 1-based values as in the reference
        self.JYO = np.where(north[:, None], [1, 2], [2, 1]).astype(np.int32)
        self.JXO = np.where(east[:, None], [1, 2], [2, 1]).astype(np.int32)
        kcr = np.zeros((NANG, 4), dtype=np.int32)
        kcr[north & east] = (3, 2, 4, 1)
        kcr[north & ~east] = (2, 3, 1, 4)
        kcr[~north & east] = (4, 1, 3, 2)
        kcr[~north & ~east] = (1, 4, 2, 3)
        self.KCR = kcr
