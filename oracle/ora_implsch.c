/*
 * oracle/ora_implsch.c -- TEST INFRASTRUCTURE ONLY. See ora.h header ("parity unpinned").
 * Plain-C restatement of the IMPLSCH call tree (implsch.F90 + the routines it inlines),
 * one sea point at a time (the reference's IJ loop is innermost and couples no points).
 * F(k,m) below is FL1(IJ,K,M) of the reference with 0-based k,m.
 */
#include <float.h>
#include "ora.h"

#define NA ORA_MAXANG
#define NF ORA_MAXFRE

typedef struct {
  /* in */
  const real *WAVNUM, *CGROUP, *CINV, *XK2CG, *STOKFAC; /* [NFRE] */
  real EMAXDPT, DEPTH;
  int IOBND, IODP;
  /* inout forcing */
  real AIRD, WDWAVE, CICOVER, WSWAVE, WSTAR, USTRA, VSTRA;
  real UFRIC, TAUW, TAUWDIR, Z0M, Z0B, CHRNCK, CITHICK;
  /* inout integrated fields */
  real IBRMEM; /* ENVIRONMENT%IBRMEM (in) */
  real WSEMEAN, WSFMEAN, USTOKES, VSTOKES, STRNMS;
  real TAUXD, TAUYD, TAUOCXD, TAUOCYD, TAUOC, TAUICX, TAUICY, PHIOCD, PHIEPS, PHIAW;
  /* NEMO (JWRO = double) */
  double NEMOUSTOKES, NEMOVSTOKES, NEMOSTRN, NPHIEPS, NTAUOC, NSWH, NMWP, NEMOTAUX, NEMOTAUY;
  double NEMOTAUICX, NEMOTAUICY, NEMOWSWAVE, NEMOPHIF;
  int MIJ; /* 1-based like the reference */
} point_t;

#define F(k, m) FL1[(k) * NFRE + (m)]
#define X3(A, k, m) A[(k) * NFRE + (m)]

/* semean.F90:82-120 */
static real semean(const real *FL1, int LLEPSMIN) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real EM = LLEPSMIN ? S.EPSMIN : C_(0.0), TEMP = 0;
  for (int M = 0; M < NFRE; M++) {
    TEMP = F(0, M);
    for (int K = 1; K < NANG; K++) TEMP = TEMP + F(K, M);
    EM = EM + S.DFIM[M] * TEMP;
  }
  real DELT25 = S.WETAIL * S.FR[NFRE - 1] * S.DELTH;
  EM = EM + DELT25 * TEMP;
  return EM;
}

/* sdepthlim.F90:64-78 */
static void sdepthlim(real EMAXDPT, real *FL1) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real EM = semean(FL1, 1);
  EM = RMIN(EMAXDPT / EM, C_(1.0));
  for (int M = 0; M < NFRE; M++)
    for (int K = 0; K < NANG; K++) F(K, M) = RMAX(F(K, M) * EM, S.EPSMIN);
}

/* fkmean.F90:94-150 */
static void fkmean(const real *FL1, const real *WAVNUM, real *EM, real *FM1, real *F1, real *AK, real *XK) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real DELT25, COEFM1, COEF1, COEFA, COEFX, SQRTK, TEMPA, TEMPX, TEMP2 = 0;
  *EM = S.EPSMIN; *FM1 = S.EPSMIN; *F1 = S.EPSMIN; *AK = S.EPSMIN; *XK = S.EPSMIN;
  DELT25 = S.WETAIL * S.FR[NFRE - 1] * S.DELTH;
  COEFM1 = S.FRTAIL * S.DELTH;
  COEF1 = S.WP1TAIL * S.DELTH * S.FR[NFRE - 1] * S.FR[NFRE - 1];
  COEFA = COEFM1 * SQRT(S.G) / S.ZPI;
  COEFX = COEF1 * (S.ZPI / SQRT(S.G));
  for (int M = 0; M < NFRE; M++) {
    SQRTK = SQRT(WAVNUM[M]);
    TEMPA = S.DFIM[M] / SQRTK;
    TEMPX = SQRTK * S.DFIM[M];
    TEMP2 = F(0, M);
    for (int K = 1; K < NANG; K++) TEMP2 = TEMP2 + F(K, M);
    *EM = *EM + S.DFIM[M] * TEMP2;
    *FM1 = *FM1 + S.DFIMOFR[M] * TEMP2;
    *F1 = *F1 + S.DFIMFR[M] * TEMP2;
    *AK = *AK + TEMPA * TEMP2;
    *XK = *XK + TEMPX * TEMP2;
  }
  *EM = *EM + DELT25 * TEMP2;
  *FM1 = *FM1 + COEFM1 * TEMP2;
  *FM1 = *EM / *FM1;
  *F1 = *F1 + COEF1 * TEMP2;
  *F1 = *F1 / *EM;
  *AK = *AK + COEFA * TEMP2;
  *AK = (*EM / *AK) * (*EM / *AK);
  *XK = *XK + COEFX * TEMP2;
  *XK = (*XK / *EM) * (*XK / *EM);
}

/* femeanws.F90:84-123 */
static void femeanws(const real *FL1, const real *XLLWS, real *FM, real *EM) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real EM_LOC = S.EPSMIN, TEMP2 = 0;
  *FM = S.EPSMIN;
  real DELT25 = S.WETAIL * S.FR[NFRE - 1] * S.DELTH;
  real DELT2 = S.FRTAIL * S.DELTH;
  for (int M = 0; M < NFRE; M++) {
    TEMP2 = C_(0.0);
    for (int K = 0; K < NANG; K++) TEMP2 = TEMP2 + X3(XLLWS, K, M) * F(K, M);
    EM_LOC = EM_LOC + S.DFIM[M] * TEMP2;
    *FM = *FM + S.DFIMOFR[M] * TEMP2;
  }
  EM_LOC = EM_LOC + DELT25 * TEMP2;
  *FM = *FM + DELT2 * TEMP2;
  *FM = EM_LOC / *FM;
  if (EM) *EM = EM_LOC;
}

/* chnkmin.F90:58 */
static real chnkmin(real U10) {
  return S.ALPHAMIN + (S.ALPHA - S.ALPHAMIN) * C_(0.5) * (C_(1.0) - TANH(U10 - S.CHNKMIN_U));
}
/* cdm.func.h */
static real cdm(real U) {
  return RMAX(RMIN(C_(0.0006) + C_(0.00008) * U, C_(0.001) + C_(0.0018) * EXP(-C_(0.05) * (U - C_(33.)))), C_(0.001));
}
/* ns_gc.F90:47-49 */
static int ns_gc(real USTAR) {
  real XKS = S.SQRTGOSURFT / (C_(1.48) + C_(2.05) * USTAR);
  int n = (int)(LOG(RMAX(XKS * S.XKM_GC[1], C_(1.0))) * S.XLOGKRATIOM1_GC) + 1;
  return n < S.NWAV_GC - 1 ? n : S.NWAV_GC - 1;
}
/* stress_gc.F90:80-130 */
static real stress_gc(real ANG_GC, real USTAR, real Z0, real Z0MIN, real HALP, real RNFAC) {
  const real XLAMA = C_(0.25), XLAMB = C_(4.0);
  int NS = ns_gc(USTAR);
  real TAUWCG_MIN, TAUWCG, ZABHRC, X, XLOG, ZLOG, ZLOG2X, CONST, ZN, GAMNORMA, GAM_W, XLAMBDA;
  real t = USTAR * (Z0MIN / Z0);
  TAUWCG_MIN = t * t;
  XLAMBDA = C_(1.0) + XLAMA * TANH(XLAMB * powi(USTAR, 4));
  ZABHRC = ANG_GC * S.BETAMAXOXKAPPA2 * HALP * S.C2OSQRTVG_GC[NS];
  if (S.c.llnormagam) CONST = RNFAC * S.BMAXOKAP * HALP * S.C2OSQRTVG_GC[NS] / RMAX(USTAR, S.EPSUS);
  else CONST = C_(0.0);
  X = USTAR * S.CM_GC[NS];
  XLOG = LOG(S.XK_GC[NS] * Z0) + S.XKAPPA / (X + S.ZALP);
  ZLOG = XLOG - LOG(XLAMBDA);
  ZLOG = RMIN(ZLOG, C_(0.0));
  ZLOG2X = ZLOG * ZLOG * X;
  GAM_W = ZLOG2X * ZLOG2X * EXP(XLOG) * S.OM3GMKM_GC[NS];
  ZN = CONST * S.XKMSQRTVGOC2_GC[NS] * GAM_W;
  GAMNORMA = (C_(1.0) + S.RN1_RN * ZN) / (C_(1.0) + ZN);
  TAUWCG = GAM_W * S.DELKCC_GC_NS[NS] * S.OMXKM3_GC[NS] * GAMNORMA;
  for (int I = NS + 1; I <= S.NWAV_GC; I++) {
    X = USTAR * S.CM_GC[I];
    XLOG = LOG(S.XK_GC[I] * Z0) + S.XKAPPA / (X + S.ZALP);
    ZLOG = XLOG - LOG(XLAMBDA);
    ZLOG = RMIN(ZLOG, C_(0.0));
    ZLOG2X = ZLOG * ZLOG * X;
    GAM_W = ZLOG2X * ZLOG2X * EXP(XLOG) * S.OM3GMKM_GC[I];
    ZN = CONST * S.XKMSQRTVGOC2_GC[I] * GAM_W;
    GAMNORMA = (C_(1.0) + S.RN1_RN * ZN) / (C_(1.0) + ZN);
    TAUWCG = TAUWCG + GAM_W * S.DELKCC_OMXKM3_GC[I] * GAMNORMA;
  }
  return RMAX(ZABHRC * TAUWCG, TAUWCG_MIN);
}

/* taut_z0.F90:132-340 */
static void taut_z0(int IUSFG, real HALP, real UTOP, real UDIR, real TAUW, real TAUWDIR, real RNFAC,
                    real *USTAR, real *Z0, real *Z0B, real *CHRNCK) {
  const int NITER = 18;
  const real TWOXMP1 = C_(3.0), PMAX = C_(0.99), Z0MIN = C_(0.000001);
  int ITER;
  real ALPHAGM1, PCE_GC, Z0MINRST, CHARNOCK_MIN, COSDIFF, ZCHAR, US2TOTAUW, USMAX, XLOGXL, XKUTOP, XOLOGZ0;
  real USTOLD, USTNEW, TAUOLD, TAUNEW, X, Fv, DELF, CDFG, USNRF, Z0NRF, Z0BNRF, ALPOG;
  real USTM1, Z0TOT, Z0CH = 0, Z0VIS, HZ0VISO1MX, ZZ, TAUV, DEL, RNUEFF, RNUKAPPAM1;
  real ALPHAOG, XMIN, W1, TAUWACT, TAUWEFF, ANG_GC, TAUUNR = 0;
  int LLCOSDIFF;

  XLOGXL = LOG(S.XNLEV);
  US2TOTAUW = C_(1.0) + S.EPS1;
  RNUEFF = C_(0.04) * S.RNU;
  RNUKAPPAM1 = RNUEFF / S.XKAPPA;
  PCE_GC = C_(0.001) * IUSFG + (1 - IUSFG) * C_(0.005);

  COSDIFF = COS(UDIR - TAUWDIR);
  TAUWACT = RMAX(TAUW * COSDIFF, S.EPSMIN);
  LLCOSDIFF = (COSDIFF > C_(0.9));

  if (S.c.llgcbz0) {
    if (S.c.llcapchnk) { CHARNOCK_MIN = chnkmin(UTOP); ALPHAOG = CHARNOCK_MIN * S.GM1; }
    else ALPHAOG = C_(0.0);
    USMAX = RMAX(-C_(0.21339) + C_(0.093698) * UTOP - C_(0.0020944) * UTOP * UTOP + C_(5.5091E-5) * UTOP * UTOP * UTOP, C_(0.03));
    TAUWEFF = RMIN(TAUWACT * US2TOTAUW, USMAX * USMAX);
    if (IUSFG == 0) {
      ALPHAGM1 = S.ALPHA * S.GM1;
      if (UTOP < C_(1.0)) CDFG = C_(0.002);
      else if (LLCOSDIFF) {
        real um = RMAX(*USTAR, S.EPSUS);
        X = RMIN(TAUWACT / (um * um), PMAX);
        ZCHAR = RMIN(ALPHAGM1 * (*USTAR) * (*USTAR) / SQRT(C_(1.0) - X), C_(0.05) * EXP(-C_(0.05) * (UTOP - C_(35.))));
        ZCHAR = RMIN(ZCHAR, S.ALPHAMAX);
        CDFG = S.ACDLIN + S.BCDLIN * SQRT(ZCHAR) * UTOP;
      } else CDFG = cdm(UTOP);
      *USTAR = UTOP * SQRT(CDFG);
    }
    W1 = C_(0.85) - C_(0.05) * (TANH(C_(10.0) * (UTOP - C_(5.0))) + C_(1.0));
    XKUTOP = S.XKAPPA * UTOP;
    USTOLD = *USTAR;
    TAUOLD = USTOLD * USTOLD;
    for (ITER = 1; ITER <= NITER; ITER++) {
      *Z0 = RMAX(S.XNLEV / (EXP(RMIN(XKUTOP / USTOLD, C_(50.0))) - C_(1.0)), Z0MIN);
      TAUV = RNUKAPPAM1 * USTOLD / *Z0;
      ANG_GC = S.ANG_GC_A + S.ANG_GC_B * TANH(S.ANG_GC_C * TAUOLD);
      TAUUNR = stress_gc(ANG_GC, *USTAR, *Z0, Z0MIN, HALP, RNFAC);
      TAUNEW = TAUWEFF + TAUV + TAUUNR;
      USTNEW = SQRT(TAUNEW);
      *USTAR = W1 * USTOLD + (C_(1.0) - W1) * USTNEW;
      DEL = *USTAR - USTOLD;
      if (FABS(DEL) < PCE_GC * (*USTAR)) break;
      TAUOLD = (*USTAR) * (*USTAR);
      USTOLD = *USTAR;
    }
    X = TAUWEFF / TAUOLD;
    if (ITER > NITER && X >= PMAX) {
      CDFG = cdm(UTOP);
      *USTAR = UTOP * SQRT(CDFG);
      Z0MINRST = (*USTAR) * (*USTAR) * S.ALPHA * S.GM1;
      *Z0 = RMAX(S.XNLEV / (EXP(XKUTOP / *USTAR) - C_(1.0)), Z0MINRST);
      *Z0B = Z0MINRST;
    } else {
      *Z0 = RMAX(S.XNLEV / (EXP(XKUTOP / *USTAR) - C_(1.0)), Z0MIN);
      *Z0B = *Z0 * SQRT(TAUUNR / TAUOLD);
    }
    if (X < PMAX) {
      USNRF = *USTAR; Z0NRF = *Z0; Z0BNRF = *Z0B;
      USTOLD = *USTAR;
      TAUOLD = RMAX(USTOLD * USTOLD, TAUWEFF);
      ALPOG = RMAX(RMIN(*Z0B / TAUOLD, S.ALPHAMAX), ALPHAOG);
      for (ITER = 1; ITER <= NITER; ITER++) {
        X = RMIN(TAUWEFF / TAUOLD, PMAX);
        USTM1 = C_(1.0) / RMAX(USTOLD, S.EPSUS);
        Z0VIS = S.RNUM * USTM1;
        HZ0VISO1MX = C_(0.5) * Z0VIS / (C_(1.0) - X);
        *Z0B = ALPOG * TAUOLD;
        *Z0 = HZ0VISO1MX + SQRT(HZ0VISO1MX * HZ0VISO1MX + (*Z0B) * (*Z0B) / (C_(1.0) - X));
        XOLOGZ0 = C_(1.0) / LOG(S.XNLEV / *Z0 + C_(1.0));
        Fv = USTOLD - XKUTOP * XOLOGZ0;
        ZZ = C_(2.0) * USTM1 * (C_(3.0) * (*Z0B) * (*Z0B) + C_(0.5) * Z0VIS * (*Z0) - (*Z0) * (*Z0)) /
             (C_(2.0) * (*Z0) * (*Z0) * (C_(1.0) - X) - Z0VIS * (*Z0));
        DELF = C_(1.0) - XKUTOP * XOLOGZ0 * XOLOGZ0 * ZZ;
        if (DELF != C_(0.0)) *USTAR = USTOLD - Fv / DELF;
        TAUNEW = RMAX((*USTAR) * (*USTAR), TAUWEFF);
        *USTAR = SQRT(TAUNEW);
        DEL = TAUNEW - TAUOLD;
        if (FABS(DEL) < PCE_GC * TAUOLD) break;
        TAUOLD = TAUNEW;
        USTOLD = *USTAR;
      }
      if (ITER > NITER) {
        *USTAR = USNRF; *Z0 = Z0NRF; *Z0B = Z0BNRF;
        USTM1 = C_(1.0) / RMAX(*USTAR, S.EPSUS);
        Z0VIS = S.RNUM * USTM1;
        *CHRNCK = RMAX(S.G * (*Z0 - Z0VIS) * USTM1 * USTM1, S.ALPHAMIN);
      } else {
        real um = RMAX(*USTAR, S.EPSUS);
        *CHRNCK = RMAX(S.G * (*Z0B / SQRT(C_(1.0) - X)) / (um * um), S.ALPHAMIN);
      }
    } else {
      USTM1 = C_(1.0) / RMAX(*USTAR, S.EPSUS);
      Z0VIS = S.RNUM * USTM1;
      *CHRNCK = RMAX(S.G * (*Z0 - Z0VIS) * USTM1 * USTM1, S.ALPHAMIN);
    }
  } else {
    TAUWEFF = TAUWACT * US2TOTAUW;
    if (S.c.llcapchnk) {
      CHARNOCK_MIN = chnkmin(UTOP);
      XMIN = C_(0.15) * (S.ALPHA - CHARNOCK_MIN);
      ALPHAOG = CHARNOCK_MIN * S.GM1;
    } else { XMIN = C_(0.0); ALPHAOG = S.ALPHA * S.GM1; }
    XKUTOP = S.XKAPPA * UTOP;
    USTOLD = (1 - IUSFG) * UTOP * SQRT(RMIN(S.ACD + S.BCD * UTOP, S.CDMAX)) + IUSFG * (*USTAR);
    TAUOLD = RMAX(USTOLD * USTOLD, TAUWEFF);
    *USTAR = SQRT(TAUOLD);
    USTM1 = C_(1.0) / RMAX(*USTAR, S.EPSUS);
    for (ITER = 1; ITER <= NITER; ITER++) {
      X = RMAX(TAUWACT / TAUOLD, XMIN);
      Z0CH = ALPHAOG * TAUOLD / SQRT(C_(1.0) - X);
      Z0VIS = S.RNUM * USTM1;
      Z0TOT = Z0CH + Z0VIS;
      XOLOGZ0 = C_(1.0) / (XLOGXL - LOG(Z0TOT));
      Fv = *USTAR - XKUTOP * XOLOGZ0;
      ZZ = USTM1 * (Z0CH * (C_(2.0) - TWOXMP1 * X) / (C_(1.0) - X) - Z0VIS) / Z0TOT;
      DELF = C_(1.0) - XKUTOP * XOLOGZ0 * XOLOGZ0 * ZZ;
      if (DELF != C_(0.0)) *USTAR = *USTAR - Fv / DELF;
      TAUNEW = RMAX((*USTAR) * (*USTAR), TAUWEFF);
      *USTAR = SQRT(TAUNEW);
      if (TAUNEW == TAUOLD) break;
      USTM1 = C_(1.0) / RMAX(*USTAR, S.EPSUS);
      TAUOLD = TAUNEW;
    }
    *Z0 = Z0CH;
    *Z0B = ALPHAOG * TAUOLD;
    *CHRNCK = RMAX(S.G * (*Z0) * USTM1 * USTM1, S.ALPHAMIN);
  }
}

/* z0wave.F90:73-92: roughness from the friction velocity and the wave stress */
static void z0wave(real US, real TAUW, real UTOP, real *Z0, real *Z0B, real *CHRNCK) {
  const real ALPHAOG = (S.c.llcapchnk ? chnkmin(UTOP) : S.ALPHA) * S.GM1;
  const real UST2 = powi(US, 2), UST3 = powi(US, 3);
  const real ARG = RMAX(UST2 - TAUW, S.EPS1);
  *Z0 = ALPHAOG * UST3 / SQRT(ARG);
  *Z0B = ALPHAOG * UST2;
  *CHRNCK = S.G * (*Z0) / UST2;
}
/* airsea.F90:93-127: ICODE_WND = 3 (U10 given: TAUT_Z0 gives US), 1 or 2 (US given: Z0WAVE, then U10 from the log profile) */
static int airsea(real HALP, real *U10, real U10DIR, real TAUW, real TAUWDIR, real RNFAC,
                  real *US, real *Z0, real *Z0B, real *CHRNCK, int ICODE_WND, int IUSFG) {
  if (ICODE_WND == 3) { taut_z0(IUSFG, HALP, *U10, U10DIR, TAUW, TAUWDIR, RNFAC, US, Z0, Z0B, CHRNCK); return 0; }
  if (ICODE_WND == 1 || ICODE_WND == 2) {
    z0wave(*US, TAUW, *U10, Z0, Z0B, CHRNCK);
    const real XKAPPAD = C_(1.0) / S.XKAPPA, XLOGLEV = LOG(S.XNLEV);
    *U10 = XKAPPAD * (*US) * (XLOGLEV - LOG(*Z0));
    *U10 = RMAX(*U10, S.WSPMIN);
    return 0;
  }
  return 1;
}

/* wsigstar.F90:87-129 */
static real wsigstar(real WSWAVE, real UFRIC, real Z0M, real WSTAR) {
  const real BG_GUST = C_(0.0), ONETHIRD = C_(1.0) / C_(3.0), SIG_NMAX = C_(0.9);
  const real C1 = C_(1.03E-3), C2 = C_(0.04E-3), P1 = C_(1.48), P2 = C_(-0.21);
  real ZCHAR, C_D, DC_DDU, SIG_CONV, XKAPPAD, U10, C2U10P1, U10P2, BCD_LOC, U10M1, ZN, Z0VIS;
  if (S.c.llgcbz0 || S.c.llnormagam) {
    ZN = S.RNUM;
    U10M1 = C_(1.0) / RMAX(WSWAVE, S.WSPMIN);
    Z0VIS = ZN / RMAX(UFRIC, S.EPSUS);
    ZCHAR = S.G * (Z0M - Z0VIS) / RMAX(UFRIC * UFRIC, S.EPSUS);
    ZCHAR = RMAX(RMIN(ZCHAR, S.ALPHAMAX), S.ALPHAMIN);
    BCD_LOC = S.BCDLIN * SQRT(ZCHAR);
    C_D = S.ACDLIN + BCD_LOC * WSWAVE;
    DC_DDU = BCD_LOC;
    SIG_CONV = C_(1.0) + C_(0.5) * WSWAVE / C_D * DC_DDU;
    return RMIN(SIG_NMAX, SIG_CONV * U10M1 * POW(BG_GUST * UFRIC * UFRIC * UFRIC + C_(0.5) * S.XKAPPA * WSTAR * WSTAR * WSTAR, ONETHIRD));
  } else {
    XKAPPAD = C_(1.0) / S.XKAPPA;
    U10 = UFRIC * XKAPPAD * (LOG(C_(10.0)) - LOG(Z0M));
    U10 = RMAX(U10, S.WSPMIN);
    U10M1 = C_(1.0) / U10;
    C2U10P1 = C2 * POW(U10, P1);
    U10P2 = POW(U10, P2);
    C_D = (C1 + C2U10P1) * U10P2;
    DC_DDU = (P2 * C1 + (P1 + P2) * C2U10P1) * U10P2 * U10M1;
    SIG_CONV = C_(1.0) + C_(0.5) * U10 / C_D * DC_DDU;
    return RMIN(SIG_NMAX, SIG_CONV * U10M1 * POW(BG_GUST * UFRIC * UFRIC * UFRIC + C_(0.5) * S.XKAPPA * WSTAR * WSTAR * WSTAR, ONETHIRD));
  }
}

/* sinput_ard.F90:153-520 */
/* SIG_N, TEMP2, PTURB, AIRD_PVISC of the last sinput_ard call of this thread (read by ora_sinput_ard only) */
static __thread real sinput_aux[4];
static void sinput_ard(int NGST, int LLSNEG, const real *FL1, const real *WAVNUM, const real *CINV, const real *XK2CG,
                       real WDWAVE, real WSWAVE, real UFRIC, real Z0M, const real *COSWDIF, const real *SINWDIF2,
                       real RAORW, real WSTAR, real RNFAC, real *FLD, real *SL, real *SPOS, real *XLLWS) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  int IND, IGST;
  real CONSTN, AVG_GST, ABS_TAUWSHELTER, CONST1, ZNZ, X, ZLOG, ZLOG2X, XI, DELI1, DELI2;
  real FU = 0, FUD = 0, NU_AIR = 0, SMOOTH, HFTSWELLF6, Z0TUB, FAC_NU_AIR, FACM1_NU_AIR, DELABM1;
  real TAUPX, TAUPY, DSTAB2, CONST, SIG, SIG2, COEF = 0, COEF5 = 0, DFIM_SIG2;
  real CONSTF = 0, CONST11, CONST22, Z0VIS, Z0NOZ, FWW, PVISC, PTURB = 0, ZCN, SIG_N = 0, UORBT, AORB, TEMP, RE, RE_C, ZORB;
  real CNSN, SUMF, SUMFSIN2, CSTRNFAC = 0, FLP_AVG = 0, SLP_AVG = 0, ROGOROAIR = 0, AIRD_PVISC = 0, USG2, FLP, SLP;
  real XSTRESS[2], YSTRESS[2], TAUX[2], TAUY[2], USTP[2], USTPM1[2], USDIRP[2], UCN[2], UCNZALPD[2];
  real XNGAMCONST = 0, GAMNORMA[2], DSTAB1 = 0, TEMP1 = 0, TEMP2 = 0;
  real GAM0[NA][2], DSTAB[NA][2], COSLP[NA];
  int LTAUWSHELTER;

  AVG_GST = C_(1.0) / NGST;
  CONST1 = S.BETAMAXOXKAPPA2;
  CONSTN = S.DELTH / (S.XKAPPA * S.ZPI);
  ABS_TAUWSHELTER = FABS(S.TAUWSHELTER);
  LTAUWSHELTER = (ABS_TAUWSHELTER != C_(0.0));

  if (NGST > 1) SIG_N = wsigstar(WSWAVE, UFRIC, Z0M, WSTAR);
  if (S.c.llnormagam) CSTRNFAC = CONSTN * RNFAC / RAORW;

  if (LLSNEG) {
    NU_AIR = S.RNU;
    FACM1_NU_AIR = C_(4.0) / NU_AIR;
    FAC_NU_AIR = S.RNUM;
    FU = FABS(S.SWELLF3);
    FUD = S.SWELLF2;
    DELABM1 = (real)ORA_IAB / (S.ABMAX - S.ABMIN);
    UORBT = S.EPSMIN; AORB = S.EPSMIN;
    for (int M = 0; M < NFRE; M++) {
      SIG = S.ZPIFR[M]; SIG2 = SIG * SIG;
      DFIM_SIG2 = S.DFIM[M] * SIG2;
      TEMP = F(0, M);
      for (int K = 1; K < NANG; K++) TEMP = TEMP + F(K, M);
      UORBT = UORBT + DFIM_SIG2 * TEMP;
      AORB = AORB + S.DFIM[M] * TEMP;
    }
    UORBT = C_(2.0) * SQRT(UORBT);
    AORB = C_(2.0) * SQRT(AORB);
    RE = FACM1_NU_AIR * UORBT * AORB;
    Z0VIS = FAC_NU_AIR / RMAX(UFRIC, C_(0.0001));
    Z0TUB = S.Z0RAT * RMIN(S.Z0TUBMAX, Z0M);
    Z0NOZ = RMAX(Z0VIS, Z0TUB);
    ZORB = AORB / Z0NOZ;
    XI = (LOG10(RMAX(ZORB, C_(3.0))) - S.ABMIN) * DELABM1;
    IND = (int)XI; if (IND > ORA_IAB - 1) IND = ORA_IAB - 1;
    DELI1 = RMIN(C_(1.0), XI - (real)IND);
    DELI2 = C_(1.0) - DELI1;
    FWW = S.SWELLFT[IND] * DELI2 + S.SWELLFT[IND + 1] * DELI1; /* SWELLFT(0) is out of range in the reference too when XI<1 */
    TEMP2 = FWW * UORBT;
    if (S.SWELLF6 == C_(1.0)) RE_C = S.SWELLF4;
    else { HFTSWELLF6 = C_(1.0) - S.SWELLF6; RE_C = S.SWELLF4 * POW(C_(2.0) / AORB, HFTSWELLF6); }
    if (S.SWELLF7 > C_(0.0)) {
      SMOOTH = C_(0.5) * TANH((RE - RE_C) * S.SWELLF7M1);
      PTURB = C_(0.5) + SMOOTH;
      PVISC = C_(0.5) - SMOOTH;
    } else if (RE <= RE_C) { PTURB = C_(0.0); PVISC = C_(0.5); }
    else { PTURB = C_(0.5); PVISC = C_(0.0); }
    AIRD_PVISC = PVISC * RAORW;
  }

  sinput_aux[0] = SIG_N; sinput_aux[1] = TEMP2; sinput_aux[2] = PTURB; sinput_aux[3] = AIRD_PVISC;
  if (NGST == 1) USTP[0] = UFRIC;
  else { USTP[0] = UFRIC * (C_(1.0) + SIG_N); USTP[1] = UFRIC * (C_(1.0) - SIG_N); }
  for (IGST = 0; IGST < NGST; IGST++) USTPM1[IGST] = C_(1.0) / RMAX(USTP[IGST], S.EPSUS);

  if (LTAUWSHELTER) {
    for (IGST = 0; IGST < NGST; IGST++) {
      XSTRESS[IGST] = C_(0.0); YSTRESS[IGST] = C_(0.0);
      USG2 = USTP[IGST] * USTP[IGST];
      TAUX[IGST] = USG2 * SIN(WDWAVE);
      TAUY[IGST] = USG2 * COS(WDWAVE);
    }
    ROGOROAIR = S.G / RAORW;
  } else {
    for (int K = 0; K < NANG; K++) COSLP[K] = COSWDIF[K];
  }
  if (!S.c.llnormagam) { GAMNORMA[0] = C_(1.0); GAMNORMA[1] = C_(1.0); }
  if (!LLSNEG) for (int K = 0; K < NANG; K++) { DSTAB[K][0] = C_(0.0); DSTAB[K][1] = C_(0.0); }

  for (int M = 0; M < NFRE; M++) {
    SIG = S.ZPIFR[M]; SIG2 = SIG * SIG;
    CONST = SIG * CONST1;
    if (LLSNEG) {
      COEF = -S.SWELLF * C_(16.) * SIG2 / S.G;
      COEF5 = -S.SWELLF5 * C_(2.) * SQRT(C_(2.) * NU_AIR * SIG);
    }
    if (LTAUWSHELTER) {
      for (IGST = 0; IGST < NGST; IGST++) {
        TAUPX = TAUX[IGST] - ABS_TAUWSHELTER * XSTRESS[IGST];
        TAUPY = TAUY[IGST] - ABS_TAUWSHELTER * YSTRESS[IGST];
        USDIRP[IGST] = ATAN2(TAUPX, TAUPY);
        USTP[IGST] = POW(TAUPX * TAUPX + TAUPY * TAUPY, C_(0.25));
        USTPM1[IGST] = C_(1.0) / RMAX(USTP[IGST], S.EPSUS);
      }
      CONSTF = ROGOROAIR * CINV[M] * S.DFIM[M];
    }
    for (IGST = 0; IGST < NGST; IGST++) {
      UCN[IGST] = USTP[IGST] * CINV[M];
      UCNZALPD[IGST] = S.XKAPPA / (UCN[IGST] + S.ZALP);
    }
    ZCN = LOG(WAVNUM[M] * Z0M);
    CNSN = CONST * RAORW;
    for (int K = 0; K < NANG; K++) X3(XLLWS, K, M) = C_(0.0);
    if (S.c.llnormagam) XNGAMCONST = CSTRNFAC * XK2CG[M];
    if (LLSNEG) {
      DSTAB1 = COEF5 * AIRD_PVISC * WAVNUM[M];
      TEMP1 = COEF * RAORW;
    }
    for (IGST = 0; IGST < NGST; IGST++) {
      for (int K = 0; K < NANG; K++) {
        if (LTAUWSHELTER) COSLP[K] = COS(S.TH[K] - USDIRP[IGST]);
        GAM0[K][IGST] = C_(0.0);
        if (COSLP[K] > C_(0.01)) {
          X = COSLP[K] * UCN[IGST];
          ZLOG = ZCN + UCNZALPD[IGST] / COSLP[K];
          if (ZLOG < C_(0.0)) {
            ZLOG2X = ZLOG * ZLOG * X;
            GAM0[K][IGST] = EXP(ZLOG) * ZLOG2X * ZLOG2X * CNSN;
            X3(XLLWS, K, M) = C_(1.0);
          }
        }
      }
      if (S.c.llnormagam) {
        SUMF = C_(0.0); SUMFSIN2 = C_(0.0);
        for (int K = 0; K < NANG; K++) {
          SUMF = SUMF + GAM0[K][IGST] * F(K, M);
          SUMFSIN2 = SUMFSIN2 + GAM0[K][IGST] * F(K, M) * SINWDIF2[K];
        }
        ZNZ = XNGAMCONST * USTPM1[IGST];
        GAMNORMA[IGST] = (C_(1.0) + ZNZ * SUMFSIN2) / (C_(1.0) + ZNZ * SUMF);
      }
      if (LLSNEG) {
        for (int K = 0; K < NANG; K++) {
          DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * COSLP[K]) * USTP[IGST]);
          DSTAB[K][IGST] = DSTAB1 + PTURB * DSTAB2;
        }
      }
    }
    for (int K = 0; K < NANG; K++) {
      for (IGST = 0; IGST < NGST; IGST++) {
        SLP = GAM0[K][IGST] * GAMNORMA[IGST];
        FLP = SLP + DSTAB[K][IGST];
        SLP = SLP * F(K, M);
        if (LTAUWSHELTER) {
          CONST11 = CONSTF * S.SINTH[K];
          CONST22 = CONSTF * S.COSTH[K];
          XSTRESS[IGST] = XSTRESS[IGST] + SLP * CONST11;
          YSTRESS[IGST] = YSTRESS[IGST] + SLP * CONST22;
        }
        if (IGST == 0) { SLP_AVG = SLP; FLP_AVG = FLP; }
        else { SLP_AVG = SLP_AVG + SLP; FLP_AVG = FLP_AVG + FLP; }
      }
      X3(SPOS, K, M) = AVG_GST * SLP_AVG;
      X3(FLD, K, M) = AVG_GST * FLP_AVG;
      X3(SL, K, M) = X3(FLD, K, M) * F(K, M);
    }
  }
}

/* sinput_jan.F90:171-396 (IPHYS = 0): Janssen (1991) wind input, optional gustiness (NGST = 2), growth renormalisation
 * (LLNORMAGAM) and the swell damping of IDAMPING = 1 */
static void sinput_jan(int NGST, int LLSNEG, const real *FL1, const real *WAVNUM, const real *CINV, const real *XK2CG,
                       real WSWAVE, real UFRIC, real Z0M, const real *COSWDIF, const real *SINWDIF2, real RAORW, real WSTAR,
                       real RNFAC, real *FLD, real *SL, real *SPOS, real *XLLWS) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real CONST1 = S.BETAMAXOXKAPPA2;
  real CONST3 = C_(2.0) * S.XKAPPA / CONST1;
  real XKAPPAD = C_(1.E0) / S.XKAPPA;
  CONST3 = S.IDAMPING * CONST3;
  real CONSTN = S.DELTH / (S.XKAPPA * S.ZPI);
  real SIG_N = C_(0.0), CSTRNFAC = C_(0.0), WSIN[2], SIGDEV[2], US[2], Z0[2], USTPM1[2];
  real UCN[2], CONST3_UCN2[2], UCND[2], ZCN[2], XVD[2], GAMNORMA[2];
  static __thread real GAM0[2][NA];
  int LZ[NA];
  if (NGST > 1) SIG_N = wsigstar(WSWAVE, UFRIC, Z0M, WSTAR);
  for (int K = 0; K < NANG; K++) LZ[K] = (COSWDIF[K] > C_(0.01));
  if (S.c.llnormagam) CSTRNFAC = CONSTN * RNFAC / RAORW;
  if (NGST == 1) { WSIN[0] = C_(1.0); SIGDEV[0] = C_(1.0); }
  else { WSIN[0] = C_(0.5); WSIN[1] = C_(0.5); SIGDEV[0] = C_(1.0) - SIG_N; SIGDEV[1] = C_(1.0) + SIG_N; }
  if (NGST == 1) { US[0] = UFRIC; Z0[0] = Z0M; }
  else for (int IG = 0; IG < NGST; IG++) { US[IG] = UFRIC * SIGDEV[IG]; Z0[IG] = Z0M; }
  for (int IG = 0; IG < NGST; IG++) USTPM1[IG] = C_(1.0) / RMAX(US[IG], S.EPSUS);
  for (int M = 0; M < NFRE; M++) {
    real CONST = S.ZPIFR[M] * CONST1;
    real ZTANHKD = S.ZPIFR[M] * S.ZPIFR[M] / (S.G * WAVNUM[M]);
    real CNSN = CONST * ZTANHKD * RAORW;
    for (int IG = 0; IG < NGST; IG++) {
      UCN[IG] = US[IG] * CINV[M] + S.ZALP;
      CONST3_UCN2[IG] = CONST3 * (UCN[IG] * UCN[IG]);
      UCND[IG] = C_(1.0) / UCN[IG];
      ZCN[IG] = LOG(WAVNUM[M] * Z0[IG]);
      XVD[IG] = C_(1.0) / (-US[IG] * XKAPPAD * ZCN[IG] * CINV[M]);
    }
    for (int K = 0; K < NANG; K++) {
      X3(XLLWS, K, M) = C_(0.0);
      for (int IG = 0; IG < NGST; IG++) {
        if (LZ[K]) {
          real ZLOG = ZCN[IG] + S.XKAPPA / COSWDIF[K] * UCND[IG];
          if (ZLOG < C_(0.0)) {
            real X = COSWDIF[K] * UCN[IG];
            real ZLOG2X = ZLOG * ZLOG * X;
            GAM0[IG][K] = ZLOG2X * ZLOG2X * EXP(ZLOG) * CNSN;
            X3(XLLWS, K, M) = C_(1.0);
          } else GAM0[IG][K] = C_(0.0);
        } else GAM0[IG][K] = C_(0.0);
      }
    }
    if (S.c.llnormagam) {
      real XNGAMCONST = CSTRNFAC * XK2CG[M];
      for (int IG = 0; IG < NGST; IG++) {
        real SUMF = C_(0.0), SUMFSIN2 = C_(0.0);
        for (int K = 0; K < NANG; K++) {
          SUMF = SUMF + GAM0[IG][K] * F(K, M);
          SUMFSIN2 = SUMFSIN2 + GAM0[IG][K] * F(K, M) * SINWDIF2[K];
        }
        real ZNZ = XNGAMCONST * USTPM1[IG];
        GAMNORMA[IG] = (C_(1.0) + ZNZ * SUMFSIN2) / (C_(1.0) + ZNZ * SUMF);
      }
    } else { GAMNORMA[0] = C_(1.0); GAMNORMA[1] = C_(1.0); }
    for (int K = 0; K < NANG; K++) {
      real UFAC1 = WSIN[0] * GAM0[0][K] * GAMNORMA[0], UFAC2 = C_(0.0);
      if (NGST == 2) UFAC1 = UFAC1 + WSIN[1] * GAM0[1][K] * GAMNORMA[1];
      if (LLSNEG) {
        real ZBETA = CONST3_UCN2[0] * (COSWDIF[K] - XVD[0]);
        UFAC2 = WSIN[0] * ZBETA;
        if (NGST == 2) {
          ZBETA = CONST3_UCN2[1] * (COSWDIF[K] - XVD[1]);
          UFAC2 = UFAC2 + WSIN[1] * ZBETA;
        }
      }
      X3(FLD, K, M) = UFAC1 + UFAC2 * CNSN;
      X3(SPOS, K, M) = UFAC1 * F(K, M);
      X3(SL, K, M) = X3(FLD, K, M) * F(K, M);
    }
  }
}

/* sdissip_jan.F90:92-128 (IPHYS = 0): WAM cycle 4 whitecapping */
static void sdissip_jan(const real *FL1, real *FLD, real *SL, const real *WAVNUM, real EMEAN, real F1MEAN, real XKMEAN) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real DELTA_SDISM1 = C_(1.0) - S.DELTA_SDIS;
  real CONSS = S.CDIS * S.ZPI;
  real SDS = CONSS * F1MEAN * (EMEAN * EMEAN) * powi(XKMEAN, 4);
  for (int M = 0; M < NFRE; M++) {
    real X = WAVNUM[M] / XKMEAN;
    real XK2 = WAVNUM[M] * WAVNUM[M];
    real CVIS = S.RNU * S.CDISVIS;
    real TEMP1 = SDS * X * (DELTA_SDISM1 + S.DELTA_SDIS * X) + CVIS * XK2;
    for (int K = 0; K < NANG; K++) {
      X3(FLD, K, M) = X3(FLD, K, M) + TEMP1;
      X3(SL, K, M) = X3(SL, K, M) + TEMP1 * F(K, M);
    }
  }
}

/* frcutindex.F90:84-108 */
static int frcutindex(real FM, real FMWS, real UFRIC, real CICOVER, real *RHOWGDFTH) {
  const int NFRE = S.NFRE;
  int MIJ;
  real FPMH = S.TAILFACTOR / S.FR[0];
  real FPPM = S.TAILFACTOR_PM * S.G / (S.FRIC * S.ZPIFR[0]);
  if (CICOVER <= S.CITHRSH_TAIL) {
    real FM2 = RMAX(FMWS, FM) * FPMH;
    real FPM = FPPM / RMAX(UFRIC, S.EPSMIN);
    real FPM4 = RMAX(FM2, FPM);
    MIJ = NINT(LOG10(FPM4) * S.FLOGSPRDM1) + 1;
    MIJ = MIJ < 1 ? 1 : MIJ; MIJ = MIJ > NFRE ? NFRE : MIJ;
  } else MIJ = NFRE;
  for (int M = 1; M <= MIJ; M++) RHOWGDFTH[M - 1] = S.RHOWG_DFIM[M - 1];
  if (MIJ != NFRE) RHOWGDFTH[MIJ - 1] = C_(0.5) * RHOWGDFTH[MIJ - 1];
  for (int M = MIJ + 1; M <= NFRE; M++) RHOWGDFTH[M - 1] = C_(0.0);
  return MIJ;
}

/* tau_phi_hf.F90:125-301 */
static void tau_phi_hf(int MIJ, int LTAUWSHELTER, real UFRIC, real Z0M, const real *FL1, real AIRD, real RNFAC,
                       const real *COSWDIF, const real *SINWDIF2, real *UST, real *TAUHF, real *PHIHF, int LLPHIHF) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real ZSUPMAX = C_(0.0);
  const int JT = ORA_JTOT_TAUHF;
  real OMEGA, OMEGACC, X0G, YC, Y, CM1, ZX, ZARG, ZLOG, ZBETA, FNC2, GAMNORMA, ZNZ, CONFG, COSW, FCOSW2;
  real OMS = 0, SQRTZ0OG, ZSUP, ZINF, DELZ, TAUL, XLOGGZ0, SQRTGZ0, USTPH = 0, CONST1, CONST2, CONSTTAU, CONSTPHI;
  real F1DCOS2, F1DCOS3, F1D, F1DSIN2;
  if (S.c.llgcbz0) { int NS = ns_gc(UFRIC); OMS = S.OMEGA_GC[NS]; } /* tau_phi_hf.F90:127 -> omegagc.F90:51-55 */
  X0G = S.X0TAUHF * S.G;
  if (LLPHIHF) USTPH = *UST;
  XLOGGZ0 = LOG(S.G * Z0M);
  OMEGACC = RMAX(S.ZPIFR[MIJ - 1], X0G / *UST);
  SQRTZ0OG = SQRT(Z0M * S.GM1);
  SQRTGZ0 = C_(1.0) / SQRTZ0OG;
  YC = OMEGACC * SQRTZ0OG;
  ZINF = LOG(YC);
  CONSTTAU = S.ZPI4GM2 * S.FR5[MIJ - 1];
  {
    int K = 0;
    COSW = RMAX(COSWDIF[K], C_(0.0));
    FCOSW2 = F(K, MIJ - 1) * COSW * COSW;
    F1DCOS3 = FCOSW2 * COSW; F1DCOS2 = FCOSW2;
    F1DSIN2 = F(K, MIJ - 1) * SINWDIF2[K];
    F1D = F(K, MIJ - 1);
    for (K = 1; K < NANG; K++) {
      COSW = RMAX(COSWDIF[K], C_(0.0));
      FCOSW2 = F(K, MIJ - 1) * COSW * COSW;
      F1DCOS3 = F1DCOS3 + FCOSW2 * COSW;
      F1DCOS2 = F1DCOS2 + FCOSW2;
      F1DSIN2 = F1DSIN2 + F(K, MIJ - 1) * SINWDIF2[K];
      F1D = F1D + F(K, MIJ - 1);
    }
  }
  F1DCOS3 = S.DELTH * F1DCOS3; F1DCOS2 = S.DELTH * F1DCOS2; F1DSIN2 = S.DELTH * F1DSIN2; F1D = S.DELTH * F1D;
  if (S.c.llnormagam) {
    CONFG = S.GAMNCONST * S.FR5[MIJ - 1] * RNFAC * SQRTGZ0;
    CONST1 = CONFG * F1DSIN2; CONST2 = CONFG * F1D;
  } else { CONST1 = C_(0.0); CONST2 = C_(0.0); }
  if (S.c.llgcbz0) ZSUP = RMIN(LOG(OMS * SQRTZ0OG), ZSUPMAX);
  else ZSUP = ZSUPMAX;
  TAUL = (*UST) * (*UST);
  DELZ = RMAX((ZSUP - ZINF) / (real)(JT - 1), C_(0.0));
  *TAUHF = C_(0.0);
  if (LTAUWSHELTER) {
    for (int J = 1; J <= JT; J++) {
      Y = EXP(ZINF + (real)(J - 1) * DELZ);
      OMEGA = Y * SQRTGZ0;
      CM1 = OMEGA * S.GM1;
      ZX = (*UST) * CM1 + S.ZALP;
      ZARG = S.XKAPPA / ZX;
      ZLOG = XLOGGZ0 + C_(2.0) * LOG(CM1) + ZARG;
      ZLOG = RMIN(ZLOG, C_(0.0));
      ZBETA = powi(ZLOG, 4) * EXP(ZLOG);
      ZNZ = ZBETA * (*UST) * Y;
      GAMNORMA = (C_(1.0) + CONST1 * ZNZ) / (C_(1.0) + CONST2 * ZNZ);
      FNC2 = F1DCOS3 * CONSTTAU * ZBETA * TAUL * S.WTAUHF[J - 1] * DELZ * GAMNORMA;
      TAUL = RMAX(TAUL - S.TAUWSHELTER * FNC2, C_(0.0));
      *UST = SQRT(TAUL);
      *TAUHF = *TAUHF + FNC2;
    }
  } else {
    for (int J = 1; J <= JT; J++) {
      Y = EXP(ZINF + (real)(J - 1) * DELZ);
      OMEGA = Y * SQRTGZ0;
      CM1 = OMEGA * S.GM1;
      ZX = (*UST) * CM1 + S.ZALP;
      ZARG = S.XKAPPA / ZX;
      ZLOG = XLOGGZ0 + C_(2.0) * LOG(CM1) + ZARG;
      ZLOG = RMIN(ZLOG, C_(0.0));
      ZBETA = powi(ZLOG, 4) * EXP(ZLOG);
      FNC2 = ZBETA * S.WTAUHF[J - 1];
      ZNZ = ZBETA * (*UST) * Y;
      GAMNORMA = (C_(1.0) + CONST1 * ZNZ) / (C_(1.0) + CONST2 * ZNZ);
      *TAUHF = *TAUHF + FNC2 * GAMNORMA;
    }
    *TAUHF = F1DCOS3 * CONSTTAU * TAUL * (*TAUHF) * DELZ;
  }
  *PHIHF = C_(0.0);
  if (LLPHIHF) {
    TAUL = USTPH * USTPH;
    ZSUP = ZSUPMAX;
    DELZ = RMAX((ZSUP - ZINF) / (real)(JT - 1), C_(0.0));
    CONSTPHI = AIRD * S.ZPI4GM1 * S.FR5[MIJ - 1];
    if (LTAUWSHELTER) {
      for (int J = 1; J <= JT; J++) {
        Y = EXP(ZINF + (real)(J - 1) * DELZ);
        OMEGA = Y * SQRTGZ0;
        CM1 = OMEGA * S.GM1;
        ZX = USTPH * CM1 + S.ZALP;
        ZARG = S.XKAPPA / ZX;
        ZLOG = XLOGGZ0 + C_(2.0) * LOG(CM1) + ZARG;
        ZLOG = RMIN(ZLOG, C_(0.0));
        ZBETA = powi(ZLOG, 4) * EXP(ZLOG);
        ZNZ = ZBETA * (*UST) * Y;
        GAMNORMA = (C_(1.0) + CONST1 * ZNZ) / (C_(1.0) + CONST2 * ZNZ);
        FNC2 = ZBETA * TAUL * S.WTAUHF[J - 1] * DELZ * GAMNORMA;
        TAUL = RMAX(TAUL - S.TAUWSHELTER * F1DCOS3 * CONSTTAU * FNC2, C_(0.0));
        USTPH = SQRT(TAUL);
        *PHIHF = *PHIHF + FNC2 / Y;
      }
      *PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * (*PHIHF);
    } else {
      for (int J = 1; J <= JT; J++) {
        Y = EXP(ZINF + (real)(J - 1) * DELZ);
        OMEGA = Y * SQRTGZ0;
        CM1 = OMEGA * S.GM1;
        ZX = USTPH * CM1 + S.ZALP;
        ZARG = S.XKAPPA / ZX;
        ZLOG = XLOGGZ0 + C_(2.0) * LOG(CM1) + ZARG;
        ZLOG = RMIN(ZLOG, C_(0.0));
        ZBETA = powi(ZLOG, 4) * EXP(ZLOG);
        ZNZ = ZBETA * (*UST) * Y;
        GAMNORMA = (C_(1.0) + CONST1 * ZNZ) / (C_(1.0) + CONST2 * ZNZ);
        FNC2 = ZBETA * S.WTAUHF[J - 1] * GAMNORMA;
        *PHIHF = *PHIHF + FNC2 / Y;
      }
      *PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * TAUL * (*PHIHF) * DELZ;
    }
  }
}

/* stresso.F90:125-229 */
static void stresso(int MIJ, const real *RHOWGDFTH, const real *FL1, const real *SL, const real *SPOS, const real *CINV,
                    real WDWAVE, real UFRIC, real Z0M, real AIRD, real RNFAC, const real *COSWDIF, const real *SINWDIF2,
                    real *TAUW, real *TAUWDIR, real *PHIWA, int LLPHIWA) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real TAUTOUS2, XSTRESS, YSTRESS, TAUHF, PHIHF, CMRHOWGDFTH, TAUX, TAUY, TAUPX, TAUPY, USDIRP, UST, SUMT, SUMX, SUMY;
  int LTAUWSHELTER;
  *PHIWA = C_(0.0); XSTRESS = C_(0.0); YSTRESS = C_(0.0);
  if (LLPHIWA) {
    for (int M = 0; M < NFRE; M++)
      for (int K = 0; K < NANG; K++) *PHIWA = *PHIWA + (X3(SL, K, M) - X3(SPOS, K, M)) * S.RHOWG_DFIM[M];
  }
  for (int M = 0; M < NFRE; M++) {
    SUMX = X3(SPOS, 0, M) * S.SINTH[0];
    SUMY = X3(SPOS, 0, M) * S.COSTH[0];
    SUMT = X3(SPOS, 0, M);
    for (int K = 1; K < NANG; K++) {
      SUMX = SUMX + X3(SPOS, K, M) * S.SINTH[K];
      SUMY = SUMY + X3(SPOS, K, M) * S.COSTH[K];
      SUMT = SUMT + X3(SPOS, K, M);
    }
    CMRHOWGDFTH = RHOWGDFTH[M] * CINV[M];
    XSTRESS = XSTRESS + CMRHOWGDFTH * SUMX;
    YSTRESS = YSTRESS + CMRHOWGDFTH * SUMY;
    if (LLPHIWA) *PHIWA = *PHIWA + RHOWGDFTH[M] * SUMT;
  }
  XSTRESS = XSTRESS / RMAX(AIRD, C_(1.0));
  YSTRESS = YSTRESS / RMAX(AIRD, C_(1.0));
  if (S.c.iphys == 0 || S.TAUWSHELTER == C_(0.0)) {
    LTAUWSHELTER = 0; USDIRP = WDWAVE; UST = UFRIC;
  } else {
    LTAUWSHELTER = 1;
    TAUX = UFRIC * UFRIC * SIN(WDWAVE);
    TAUY = UFRIC * UFRIC * COS(WDWAVE);
    TAUPX = TAUX - S.TAUWSHELTER * XSTRESS;
    TAUPY = TAUY - S.TAUWSHELTER * YSTRESS;
    USDIRP = ATAN2(TAUPX, TAUPY);
    UST = POW(TAUPX * TAUPX + TAUPY * TAUPY, C_(0.25));
  }
  tau_phi_hf(MIJ, LTAUWSHELTER, UFRIC, Z0M, FL1, AIRD, RNFAC, COSWDIF, SINWDIF2, &UST, &TAUHF, &PHIHF, LLPHIWA);
  XSTRESS = XSTRESS + TAUHF * SIN(USDIRP);
  YSTRESS = YSTRESS + TAUHF * COS(USDIRP);
  *TAUW = SQRT(XSTRESS * XSTRESS + YSTRESS * YSTRESS);
  *TAUW = RMAX(*TAUW, C_(0.0));
  *TAUWDIR = ATAN2(XSTRESS, YSTRESS);
  if (!S.c.llgcbz0) {
    TAUTOUS2 = C_(1.0) / (C_(1.0) + S.EPS1);
    *TAUW = RMIN(*TAUW, UFRIC * UFRIC * TAUTOUS2);
  }
  if (LLPHIWA) *PHIWA = *PHIWA + PHIHF;
}

/* halphap.F90:68-112 with meansqs_lf.F90:80-100 and femean.F90:84-121 */
static real halphap(const real *WAVNUM, const real *COSWDIF, const real *FL1) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  static __thread real FLWD[NA * NF];
  real WD[NA], XMSS, EM, FM, F1D, ALPHAP, TEMP2 = 0;
  real ZLNFRNFRE = LOG(S.FR[NFRE - 1]);
  for (int K = 0; K < NANG; K++) WD[K] = C_(0.5) + C_(0.5) * SIGN(C_(1.0), COSWDIF[K]);
  for (int M = 0; M < NFRE; M++)
    for (int K = 0; K < NANG; K++) X3(FLWD, K, M) = F(K, M) * WD[K];
  XMSS = C_(0.0);
  for (int M = 0; M < NFRE; M++) {
    real TEMP1 = S.DFIM[M] * WAVNUM[M] * WAVNUM[M];
    TEMP2 = C_(0.0);
    for (int K = 0; K < NANG; K++) TEMP2 = TEMP2 + X3(FLWD, K, M);
    XMSS = XMSS + TEMP1 * TEMP2;
  }
  EM = C_(0.0); FM = C_(0.0);
  for (int M = 0; M < NFRE; M++) {
    TEMP2 = RMAX(X3(FLWD, 0, M), S.EPSMIN);
    for (int K = 1; K < NANG; K++) TEMP2 = TEMP2 + RMAX(X3(FLWD, K, M), S.EPSMIN);
    EM = EM + TEMP2 * S.DFIM[M];
    FM = FM + S.DFIMOFR[M] * TEMP2;
  }
  EM = EM + S.WETAIL * S.FR[NFRE - 1] * S.DELTH * TEMP2;
  FM = FM + S.FRTAIL * S.DELTH * TEMP2;
  FM = EM / FM;
  FM = RMAX(FM, S.FR[0]);
  if (EM > C_(0.0) && FM < S.FR[NFRE - 3]) {
    ALPHAP = XMSS / (ZLNFRNFRE - LOG(FM));
    if (ALPHAP > S.ALPHAPMAX) {
      F1D = C_(0.0);
      for (int K = 0; K < NANG; K++) F1D = F1D + X3(FLWD, K, NFRE - 1) * S.DELTH;
      ALPHAP = S.ZPI4GM2 * S.FR5[NFRE - 1] * F1D;
    }
  } else {
    F1D = C_(0.0);
    for (int K = 0; K < NANG; K++) F1D = F1D + X3(FLWD, K, NFRE - 1) * S.DELTH;
    ALPHAP = S.ZPI4GM2 * S.FR5[NFRE - 1] * F1D;
  }
  return C_(0.5) * RMIN(ALPHAP, S.ALPHAPMAX);
}

/* sinflx.F90:105-183 */
static int sinflx(int ICALL, int NCALL, int LUPDTUS, real *FL1, point_t *p, real RAORW, const real *COSWDIF,
                  const real *SINWDIF2, real FMEAN, real *HALP, real *FMEANWS, const real *FLM, real *PHIWA, real *FLD,
                  real *SL, real *SPOS, real *RHOWGDFTH, real *XLLWS) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  int IUSFG, ICODE_WND, NGST, LLPHIWA, LLSNEG;
  real RNFAC;
  if (ICALL == 1) { IUSFG = 0; ICODE_WND = S.c.icode; }
  else { IUSFG = 1; ICODE_WND = 3; }
  if (S.c.llnormagam && S.c.llcapchnk) RNFAC = C_(1.0) + S.DTHRN_A * (C_(1.0) + TANH(p->WSWAVE - S.DTHRN_U));
  else RNFAC = C_(1.0);
  if (LUPDTUS) {
    if (ICALL == 1) {
      for (int K = 0; K < NANG; K++) F(K, NFRE - 1) = RMAX(F(K, NFRE - 1), FLM[K]);
      if (S.c.llgcbz0) *HALP = halphap(p->WAVNUM, COSWDIF, FL1);
      else *HALP = C_(0.0);
    }
    if (airsea(*HALP, &p->WSWAVE, p->WDWAVE, p->TAUW, p->TAUWDIR, RNFAC, &p->UFRIC, &p->Z0M, &p->Z0B, &p->CHRNCK,
               ICODE_WND, IUSFG)) return 1;
  }
  if (ICALL < NCALL) { NGST = 1; LLPHIWA = 0; LLSNEG = 0; }
  else { NGST = 2; LLPHIWA = 1; LLSNEG = 1; }
  if (S.c.iphys == 0) /* sinput.F90:102-113 */
    sinput_jan(NGST, LLSNEG, FL1, p->WAVNUM, p->CINV, p->XK2CG, p->WSWAVE, p->UFRIC, p->Z0M, COSWDIF, SINWDIF2, RAORW, p->WSTAR,
               RNFAC, FLD, SL, SPOS, XLLWS);
  else
    sinput_ard(NGST, LLSNEG, FL1, p->WAVNUM, p->CINV, p->XK2CG, p->WDWAVE, p->WSWAVE, p->UFRIC, p->Z0M, COSWDIF, SINWDIF2,
               RAORW, p->WSTAR, RNFAC, FLD, SL, SPOS, XLLWS);
  femeanws(FL1, XLLWS, FMEANWS, NULL);
  p->MIJ = frcutindex(FMEAN, *FMEANWS, p->UFRIC, p->CICOVER, RHOWGDFTH);
  stresso(p->MIJ, RHOWGDFTH, FL1, SL, SPOS, p->CINV, p->WDWAVE, p->UFRIC, p->Z0M, p->AIRD, RNFAC, COSWDIF, SINWDIF2,
          &p->TAUW, &p->TAUWDIR, PHIWA, LLPHIWA);
  return 0;
}

/* sdissip_ard.F90:117-314 (SSDSC3 = 0: cumulative term compiled out; SSDSC5 term kept) */
static void sdissip_ard(const real *FL1, real *FLD, real *SL, const real *WAVNUM, const real *XK2CG, real UFRIC,
                        const real *COSWDIF, real RAORW) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  static __thread real BTH[NA * NF], D[NA * NF];
  real BTH0[NF], FACSAT[NF];
  real TPIINV = C_(1.0) / S.ZPI;
  real TMP03 = C_(1.0) / (S.SDSBR * S.MICHE);
  real SSDSC6M1 = C_(1.) - S.SSDSC6;
  for (int M = 0; M < NFRE; M++) FACSAT[M] = WAVNUM[M] * TPIINV * XK2CG[M];
  for (int M = 0; M < NFRE; M++) {
    BTH0[M] = C_(0.0);
    for (int K = 0; K < NANG; K++) {
      real b = C_(0.0);
      for (int K2 = 0; K2 < S.NSDSNTH * 2 + 1; K2++) {
        int KK = S.INDICESSAT[K][K2];
        b = b + S.SATWEIGHTS[K][K2] * F(KK, M);
      }
      b = b * FACSAT[M];
      X3(BTH, K, M) = b;
      BTH0[M] = RMAX(BTH0[M], b);
    }
  }
  for (int M = 0; M < NFRE; M++) {
    real SSDSC2_SIG = S.SSDSC2 * S.ZPIFR[M];
    real ZCOEF = SSDSC2_SIG * S.SSDSC6;
    real ZCOEFM1 = SSDSC2_SIG * SSDSC6M1;
    for (int K = 0; K < NANG; K++) {
      real a = RMAX(C_(0.), BTH0[M] * TMP03 - S.SSDSC4);
      real b = RMAX(C_(0.), X3(BTH, K, M) * TMP03 - S.SSDSC4);
      X3(D, K, M) = ZCOEF * (a * a) + ZCOEFM1 * (b * b); /* **IPSAT, IPSAT = 2 */
    }
  }
  if (S.SSDSC5 != C_(0.0)) {
    real TMP01 = C_(2.) * S.SSDSC5 / S.G;
    real FACTURB = TMP01 * RAORW * UFRIC * UFRIC;
    for (int M = 0; M < NFRE; M++) {
      real FACWTRB = S.ZPIFR[M] * WAVNUM[M] * FACTURB;
      for (int K = 0; K < NANG; K++) X3(D, K, M) = X3(D, K, M) - FACWTRB * COSWDIF[K];
    }
  }
  for (int M = 0; M < NFRE; M++)
    for (int K = 0; K < NANG; K++) {
      X3(SL, K, M) = X3(SL, K, M) + X3(D, K, M) * F(K, M);
      X3(FLD, K, M) = X3(FLD, K, M) + X3(D, K, M);
    }
}

/* transf.F90:44-71: narrow-band shallow-water enhancement of the nonlinear transfer */
static real transf(real XK, real D) {
  const real EPS = C_(0.0001), DKMAX = C_(40.0);
  if (D < S.BATHYMAX && D > C_(0.0)) {
    real X = XK * D;
    if (X > DKMAX) return C_(1.0);
    real T_0 = TANH(X);
    real OM = SQRT(S.G * XK * T_0);
    real C_0 = OM / XK;
    real V_G;
    if (X < EPS) V_G = C_0;
    else V_G = C_(0.5) * C_0 * (C_(1.0) + C_(2.0) * X / SINH(C_(2.0) * X));
    real DV_G = powi(T_0 - X * (C_(1.0) - T_0 * T_0), 2) + C_(4.0) * (X * X) * (T_0 * T_0) * (C_(1.0) - T_0 * T_0);
    real XNL_1 = (C_(9.0) * powi(T_0, 4) - C_(10.0) * (T_0 * T_0) + C_(9.0)) / (C_(8.0) * powi(T_0, 3));
    real XNL_2 = (powi(C_(2.0) * V_G - C_(0.5) * C_0, 2) / (S.G * D - V_G * V_G) + C_(1.0)) / X;
    real XNL = XNL_1 - XNL_2;
    return (XNL * XNL) / (DV_G * powi(T_0, 8));
  }
  return C_(1.0);
}

/* peak_ang.F90:76-174: Longuet-Higgins spectral width XNU and angular width SIG_TH around the peak of the 2-D spectrum */
static void peak_ang(const real *FL1, real *XNU, real *SIG_TH) {
  const int NANG = S.NANG, NFRE = S.NFRE;
#ifdef ORA_SINGLE
  const real ZEPSILON = C_(10.) * FLT_EPSILON;
#else
  const real ZEPSILON = C_(10.) * DBL_EPSILON;
#endif
  const int NSH = 1 + (int)(LOG(C_(1.5)) / LOG(S.FRATIO));
  real SUM0 = ZEPSILON, SUM1 = C_(0.), SUM2 = C_(0.), TEMP = C_(0.);
  for (int M = 0; M < NFRE; M++) {
    TEMP = F(0, M);
    for (int K = 1; K < NANG; K++) TEMP = TEMP + F(K, M);
    SUM0 = SUM0 + TEMP * S.DFIM[M];
    SUM1 = SUM1 + TEMP * S.DFIMFR[M];
    SUM2 = SUM2 + TEMP * (S.DFIM[M] * powi(S.FR[M], 2)); /* DFIMFR2, initmdl.F90:447 */
  }
  const real DELT25 = S.WETAIL * S.FR[NFRE - 1] * S.DELTH;
  const real COEF_FR = S.WP1TAIL * S.DELTH * powi(S.FR[NFRE - 1], 2);
  const real COEF_FR2 = C_(0.5) * S.DELTH * powi(S.FR[NFRE - 1], 3); /* WP2TAIL = 0.5, yowfred.F90:54 */
  SUM0 = SUM0 + DELT25 * TEMP;
  SUM1 = SUM1 + COEF_FR * TEMP;
  SUM2 = SUM2 + COEF_FR2 * TEMP;
  if (SUM0 > ZEPSILON) *XNU = SQRT(RMAX(ZEPSILON, SUM2 * SUM0 / powi(SUM1, 2) - C_(1.)));
  else *XNU = ZEPSILON;
  real XMAX = C_(0.);
  int MMAX = 2;
  for (int M = 2; M <= NFRE - 1; M++)
    for (int K = 0; K < NANG; K++)
      if (F(K, M - 1) > XMAX) { MMAX = M; XMAX = F(K, M - 1); }
  SUM1 = ZEPSILON; SUM2 = C_(0.);
  const int MMSTART = MMAX - NSH > 1 ? MMAX - NSH : 1, MMSTOP = MMAX + NSH < NFRE ? MMAX + NSH : NFRE;
  real SUM_S = C_(0.), SUM_C = ZEPSILON;
  for (int M = MMSTART; M <= MMSTOP; M++) {
    for (int K = 0; K < NANG; K++) {
      SUM_S = SUM_S + S.SINTH[K] * F(K, M - 1);
      SUM_C = SUM_C + S.COSTH[K] * F(K, M - 1);
    }
    const real THMEAN = ATAN2(SUM_S, SUM_C);
    for (int K = 0; K < NANG; K++) {
      SUM1 = SUM1 + F(K, M - 1) * S.DFIM[M - 1];
      SUM2 = SUM2 + COS(S.TH[K] - THMEAN) * F(K, M - 1) * S.DFIM[M - 1];
    }
  }
  if (SUM1 > ZEPSILON) {
    const real R1 = SUM2 / SUM1;
    *SIG_TH = C_(1.0) * SQRT(C_(2.) * (C_(1.) - R1));
  } else *SIG_TH = C_(0.);
}

/* transf_snl.F90:52-85: shallow-water enhancement with the finite-bandwidth correction (XNU, SIG_TH from PEAK_ANG) */
static real transf_snl(real XK0, real D, real XNU, real SIG_TH) {
  const real EPS = C_(0.0001), DKMAX = C_(40.0), XKDMIN = C_(0.75); /* yowpcons.F90:34, yowshal.F90:23 */
  if (D < S.BATHYMAX && D > C_(0.)) {
    real X = XK0 * D;
    if (X > DKMAX) return C_(1.);
    const real XK = RMAX(XK0, XKDMIN / D);
    X = XK * D;
    const real T_0 = TANH(X);
    const real T_0_SQ = powi(T_0, 2);
    const real OM = SQRT(S.G * XK * T_0);
    const real C_0 = OM / XK;
    const real C_S_SQ = S.G * D;
    real V_G;
    if (X < EPS) V_G = C_0;
    else V_G = C_(0.5) * C_0 * (C_(1.) + C_(2.) * X / SINH(C_(2.) * X));
    const real V_G_SQ = powi(V_G, 2);
    const real DV_G = powi(T_0 - X * (C_(1.) - T_0_SQ), 2) + C_(4.) * powi(X, 2) * T_0_SQ * (C_(1.) - T_0_SQ);
    const real XNL_1 = (C_(9.) * powi(T_0_SQ, 2) - C_(10.) * T_0_SQ + C_(9.)) / (C_(8.) * T_0_SQ * T_0);
    const real XNL_2 = (powi(C_(2.) * V_G - C_(0.5) * C_0, 2) / (S.G * D - V_G_SQ) + C_(1.)) / X;
    const real XNL_4 = C_(1.) / (C_(4.) * T_0) * powi(C_(2.) * C_0 + V_G * (C_(1.) - T_0_SQ), 2) / (C_S_SQ - V_G_SQ);
    const real ALP = (C_(1.) - V_G_SQ / C_S_SQ) * powi(C_0, 2) / V_G_SQ;
    const real ZFAC = powi(SIG_TH, 2) / (powi(SIG_TH, 2) + ALP * powi(XNU, 2));
    const real XNL_3 = ZFAC * XNL_4;
    const real XNL = XNL_1 - XNL_2 + XNL_3;
    real r = powi(XNL, 2) / (DV_G * powi(T_0_SQ, 4));
    return RMAX(RMIN(C_(10.), r), C_(0.1));
  }
  return C_(1.);
}

/* snonlin.F90:126-494 (ISNONLIN = 0: depth scaling from AKMEAN; ISNONLIN = 1: TRANSF per interaction frequency;
 * ISNONLIN = 2: TRANSF_SNL with the spectral widths of PEAK_ANG) */
static void snonlin(const real *FL1, real *FLD, real *SL, real DEPTH, real AKMEAN, const real *WAVNUM) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real ENHFR = RMAX(C_(0.75) * DEPTH * AKMEAN, C_(0.5));
  ENHFR = C_(1.0) + (C_(5.5) / ENHFR) * (C_(1.0) - C_(.833) * ENHFR) * EXP(-C_(1.25) * ENHFR);
  int MFR1STFR = -S.MFRSTLW + 1;
  int MFRLSTFR = NFRE - S.KFRH + MFR1STFR;
  real XNU = C_(0.), SIG_TH = C_(0.);
  if (S.c.isnonlin == 2) peak_ang(FL1, &XNU, &SIG_TH);
  for (int MC = 1; MC <= S.MLSTHG; MC++) {
    int MP = S.IKP[MC - 1], MP1 = S.IKP1[MC - 1], MM = S.IKM[MC - 1], MM1 = S.IKM1[MC - 1];
    const int *I5 = S.INLCOEF[MC - 1];
    int IC = I5[0], IP = I5[1], IP1 = I5[2], IM = I5[3], IM1 = I5[4];
    const real *R = S.RNLCOEF[MC - 1];
    real FTAIL = R[0], GW1 = R[1], GW2 = R[2], GW3 = R[3], GW4 = R[4];
    real FKLAMPA = R[5], FKLAMPB = R[6], FKLAMP2 = R[7], FKLAMP1 = R[8];
    real FKLAPA2 = R[9], FKLAPB2 = R[10], FKLAP12 = R[11], FKLAP22 = R[12];
    real GW5 = R[13], GW6 = R[14], GW7 = R[15], GW8 = R[16];
    real FKLAMMA = R[17], FKLAMMB = R[18], FKLAMM2 = R[19], FKLAMM1 = R[20];
    real FKLAMA2 = R[21], FKLAMB2 = R[22], FKLAM12 = R[23], FKLAM22 = R[24];
    real ENH = ENHFR;
    if (S.c.isnonlin == 1) { /* snonlin.F90:138-150 */
      real XK = (MC <= NFRE) ? WAVNUM[MC - 1] : S.GM1 * powi(S.ZPIFR[NFRE - 1] * powi(S.FRATIO, MC - NFRE), 2);
      ENH = RMAX(RMIN(C_(10.0), transf(XK, DEPTH)), C_(0.1));
    } else if (S.c.isnonlin == 2) { /* snonlin.F90:152-165 */
      real XK = (MC <= NFRE) ? WAVNUM[MC - 1] : S.GM1 * powi(S.ZPIFR[NFRE - 1] * powi(S.FRATIO, MC - NFRE), 2);
      ENH = transf_snl(XK, DEPTH, XNU, SIG_TH);
    }
    real FTEMP = S.AF11[MC - 1] * ENH;
    int branch = (MC > MFR1STFR && MC < MFRLSTFR) ? 0 : (MC >= MFRLSTFR ? 1 : 2);
    for (int KH = 0; KH < 2; KH++) {
      for (int K = 0; K < NANG; K++) {
        int K1 = S.K1W[K][KH] - 1, K2 = S.K2W[K][KH] - 1, K11 = S.K11W[K][KH] - 1, K21 = S.K21W[K][KH] - 1;
        real SAP = GW1 * F(K1, IP - 1) + GW2 * F(K11, IP - 1) + GW3 * F(K1, IP1 - 1) + GW4 * F(K11, IP1 - 1);
        real SAM = GW5 * F(K2, IM - 1) + GW6 * F(K21, IM - 1) + GW7 * F(K2, IM1 - 1) + GW8 * F(K21, IM1 - 1);
        real FIJ = F(K, IC - 1);
        if (branch != 0) FIJ = FIJ * FTAIL;
        real FAD1 = FIJ * (SAP + SAM);
        real FAD2 = FAD1 - C_(2.0) * SAP * SAM;
        FAD1 = FAD1 + FAD2;
        real FCEN = FTEMP * FIJ;
        real AD = FAD2 * FCEN;
        real DELAD = FAD1 * FTEMP;
        real DELAP = (FIJ - C_(2.0) * SAM) * S.DAL1 * FCEN;
        real DELAM = (FIJ - C_(2.0) * SAP) * S.DAL2 * FCEN;
#define SLa(k, m1) X3(SL, k, (m1) - 1)
#define FLa(k, m1) X3(FLD, k, (m1) - 1)
        if (branch == 0) { /* :226-310 */
          SLa(K, MC) -= C_(2.0) * AD; FLa(K, MC) -= C_(2.0) * DELAD;
          SLa(K2, MM) += AD * FKLAMM1; FLa(K2, MM) += DELAM * FKLAM12;
          SLa(K21, MM) += AD * FKLAMM2; FLa(K21, MM) += DELAM * FKLAM22;
          SLa(K2, MM1) += AD * FKLAMMA; FLa(K2, MM1) += DELAM * FKLAMA2;
          SLa(K21, MM1) += AD * FKLAMMB; FLa(K21, MM1) += DELAM * FKLAMB2;
          SLa(K1, MP) += AD * FKLAMP1; FLa(K1, MP) += DELAP * FKLAP12;
          SLa(K11, MP) += AD * FKLAMP2; FLa(K11, MP) += DELAP * FKLAP22;
          SLa(K1, MP1) += AD * FKLAMPA; FLa(K1, MP1) += DELAP * FKLAPA2;
          SLa(K11, MP1) += AD * FKLAMPB; FLa(K11, MP1) += DELAP * FKLAPB2;
        } else if (branch == 1) { /* :312-412 */
          SLa(K2, MM) += AD * FKLAMM1; FLa(K2, MM) += DELAM * FKLAM12;
          SLa(K21, MM) += AD * FKLAMM2; FLa(K21, MM) += DELAM * FKLAM22;
          if (MM1 <= NFRE) {
            SLa(K2, MM1) += AD * FKLAMMA; FLa(K2, MM1) += DELAM * FKLAMA2;
            SLa(K21, MM1) += AD * FKLAMMB; FLa(K21, MM1) += DELAM * FKLAMB2;
            if (MC <= NFRE) {
              SLa(K, MC) -= C_(2.0) * AD; FLa(K, MC) -= C_(2.0) * DELAD;
              if (MP <= NFRE) {
                SLa(K1, MP) += AD * FKLAMP1; FLa(K1, MP) += DELAP * FKLAP12;
                SLa(K11, MP) += AD * FKLAMP2; FLa(K11, MP) += DELAP * FKLAP22;
                if (MP1 <= NFRE) {
                  SLa(K1, MP1) += AD * FKLAMPA; FLa(K1, MP1) += DELAP * FKLAPA2;
                  SLa(K11, MP1) += AD * FKLAMPB; FLa(K11, MP1) += DELAP * FKLAPB2;
                }
              }
            }
          }
        } else { /* :414-488 */
          if (MM1 >= 1) {
            SLa(K2, MM1) += AD * FKLAMMA; FLa(K2, MM1) += DELAM * FKLAMA2;
            SLa(K21, MM1) += AD * FKLAMMB; FLa(K21, MM1) += DELAM * FKLAMB2;
          }
          SLa(K, MC) -= C_(2.0) * AD; FLa(K, MC) -= C_(2.0) * DELAD;
          SLa(K1, MP) += AD * FKLAMP1; FLa(K1, MP) += DELAP * FKLAP12;
          SLa(K11, MP) += AD * FKLAMP2; FLa(K11, MP) += DELAP * FKLAP22;
          SLa(K1, MP1) += AD * FKLAMPA; FLa(K1, MP1) += DELAP * FKLAPA2;
          SLa(K11, MP1) += AD * FKLAMPB; FLa(K11, MP1) += DELAP * FKLAPB2;
        }
#undef SLa
#undef FLa
      }
    }
  }
}

/* sdiwbk.F90:86-117 */
static void sdiwbk(const real *FL1, real *FLD, real *SL, real DEPTH, real EMAXDPT, real EMEAN, real F1MEAN) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const float COEF_B_J = 2.0f; /* REAL default kind: 2*ALPH_B_J */
  const real DEPTHTRS = C_(50.0);
  if (!S.c.lbiwbk) return;
  if (DEPTH < DEPTHTRS) {
    real ALPH = C_(2.0) * EMAXDPT / EMEAN;
    real ARG = RMIN(ALPH, C_(50.0));
    real Q_OLD = EXP(-ARG), Q = 0, EXPQ, REL_ERR;
    for (int IC = 1; IC <= 15; IC++) {
      EXPQ = EXP(-ARG * (C_(1.0) - Q_OLD));
      Q = Q_OLD - (EXPQ - Q_OLD) / (ARG * EXPQ - C_(1.0));
      REL_ERR = FABS(Q - Q_OLD) / Q_OLD;
      if (REL_ERR < C_(0.00001)) break;
      Q_OLD = Q;
    }
    Q = RMIN(Q, C_(1.0));
    real SDS = (real)COEF_B_J * ALPH * Q * F1MEAN;
    for (int M = 0; M < S.NFRE_RED; M++)
      for (int K = 0; K < NANG; K++) {
        X3(SL, K, M) = X3(SL, K, M) - SDS * F(K, M);
        X3(FLD, K, M) = X3(FLD, K, M) - SDS;
      }
  }
}

/* sdice1.F90:104-185: scattering attenuation (Kohout & Meylan table CIDEAC, floe-size distribution of Dumont et al. 2011).
 * Every SDICEn overwrites the whole of SLICE (INTENT(OUT)): the last active one is what WNFLUXES sees. */
static void sdice1(const real *FL1, real *FLD, real *SL, real *SLICE, const real *CGROUP, real CICV, real CITH) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real DELT5 = (real)S.c.ximp * (real)S.c.idelt;
  const real CIFRGL = C_(0.955), CIDMIN = C_(20.0), CIFRGMT = C_(2.0), A = C_(200.0), C = C_(300.0);
  const int MAXICM = (int)(LOG(A / CIDMIN) / LOG(CIFRGMT));
  real DINV;
  if (CITH > C_(0.0)) {
    real CIDMAX = A + C * CICV;
    int ICM = (int)(LOG(CIDMAX / CIDMIN) / LOG(CIFRGMT));
    if (ICM > MAXICM) ICM = MAXICM;
    real SN = C_(0.0), SD = C_(0.0);
    for (int I = 0; I <= ICM; I++) {
      real X = powi(powi(CIFRGMT, 2) * CIFRGL, I);
      SN = SN + X * CIDMAX / powi(CIFRGMT, I);
      SD = SD + X;
    }
    DINV = C_(1.0) / (SN / SD);
  } else DINV = CIDMIN;
  for (int M = 0; M < NFRE; M++) {
    real ALP = C_(0.0);
    if (CITH > C_(0.0)) {
      real TW = C_(1.0) / S.FR[M];
      int IT = FLOORI((TW - S.TICMIN) / S.DTIC + 1);
      IT = IT < 1 ? 1 : (IT > S.NICT ? S.NICT : IT);
      int IT1 = IT + 1 > S.NICT ? S.NICT : IT + 1;
      real WT1 = RMAX(RMIN(C_(1.0), (TW - (S.TICMIN + (IT - 1) * S.DTIC)) / S.DTIC), C_(0.0));
      real WT = C_(1.0) - WT1;
      int IH = FLOORI((CITH - S.HICMIN) / S.DHIC + 1);
      IH = IH < 1 ? 1 : (IH > S.NICH ? S.NICH : IH);
      int IH1 = IH + 1 > S.NICH ? S.NICH : IH + 1;
      real WH1 = RMAX(RMIN(C_(1.), (CITH - (S.HICMIN + (IH - 1) * S.DHIC)) / S.DHIC), C_(0.0));
      real WH = C_(1.0) - WH1;
      real CI = WT * (WH * S.CIDEAC[IT - 1][IH - 1] + WH1 * S.CIDEAC[IT - 1][IH1 - 1]) +
                WT1 * (WH * S.CIDEAC[IT1 - 1][IH - 1] + WH1 * S.CIDEAC[IT1 - 1][IH1 - 1]);
      ALP = EXP(CI) * DINV * S.ZALPFACB;
    }
    for (int K = 0; K < NANG; K++) {
      real FLDICE = -ALP * CGROUP[M];
      X3(SLICE, K, M) = F(K, M) * FLDICE;
      X3(SL, K, M) = X3(SL, K, M) + CICV * X3(SLICE, K, M);
      X3(FLD, K, M) = X3(FLD, K, M) + CICV * FLDICE;
      real GTEMP1 = RMAX((C_(1.0) - DELT5 * FLDICE), C_(1.0));
      X3(SLICE, K, M) = X3(SLICE, K, M) / GTEMP1;
    }
  }
}
/* sdice2.F90:97-121: attenuation by ice-water drag */
static void sdice2(const real *FL1, real *FLD, real *SL, real *SLICE, const real *WAVNUM, const real *CGROUP, real CICV) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real DELT5 = (real)S.c.ximp * (real)S.c.idelt;
  for (int M = 0; M < NFRE; M++)
    for (int K = 0; K < NANG; K++) {
      real EWH = C_(4.0) * SQRT(RMAX(S.EPSMIN, F(K, M) * S.DFIM[M]));
      real XK2 = WAVNUM[M] * WAVNUM[M];
      real ALP = S.CDICWA * XK2 * EWH * S.ZALPFACB;
      real FLDICE = -ALP * CGROUP[M];
      X3(SLICE, K, M) = F(K, M) * FLDICE;
      X3(SL, K, M) = X3(SL, K, M) + CICV * X3(SLICE, K, M);
      X3(FLD, K, M) = X3(FLD, K, M) + CICV * FLDICE;
      real GTEMP1 = RMAX((C_(1.0) - DELT5 * FLDICE), C_(1.0));
      X3(SLICE, K, M) = X3(SLICE, K, M) / GTEMP1;
    }
}
/* sdice3.F90:103-160, IMODEL = 2 (Yu, Rogers & Wang 2022): viscous attenuation ~ CITH**1.25 FR**4.5 */
static void sdice3(const real *FL1, real *FLD, real *SL, real *SLICE, const real *CGROUP, real CICV, real CITH, real ALPFAC) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real DELT5 = (real)S.c.ximp * (real)S.c.idelt;
  real CDICE = C_(0.1274) * POW(S.ZPI / SQRT(S.G), C_(4.5));
  for (int M = 0; M < NFRE; M++) {
    real ALP = (C_(2.0) * CDICE * POW(CITH, C_(1.25)) * POW(S.FR[M], C_(4.5))) * ALPFAC;
    for (int K = 0; K < NANG; K++) {
      real FLDICE = -ALP * CGROUP[M];
      X3(SLICE, K, M) = F(K, M) * FLDICE;
      real TEMP = -CICV * ALP * CGROUP[M];
      X3(SL, K, M) = X3(SL, K, M) + F(K, M) * TEMP;
      X3(FLD, K, M) = X3(FLD, K, M) + TEMP;
      real GTEMP1 = RMAX((C_(1.0) - DELT5 * FLDICE), C_(1.0));
      X3(SLICE, K, M) = X3(SLICE, K, M) / GTEMP1;
    }
  }
}

/* sbottom.F90:79-97 */
static void sbottom(const real *FL1, real *FLD, real *SL, const real *WAVNUM, real DEPTH) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real CONST = -C_(2.0) * C_(0.038) * S.GM1;
  for (int M = 0; M < S.NFRE_RED; M++) {
    real SBO;
    if (DEPTH < S.BATHYMAX) {
      real ARG = C_(2.0) * DEPTH * WAVNUM[M];
      ARG = RMIN(ARG, C_(50.0));
      SBO = CONST * WAVNUM[M] / SINH(ARG);
    } else SBO = C_(0.0);
    for (int K = 0; K < NANG; K++) {
      X3(SL, K, M) = X3(SL, K, M) + SBO * F(K, M);
      X3(FLD, K, M) = X3(FLD, K, M) + SBO;
    }
  }
}

/* wnfluxes.F90:147-330 */
static void wnfluxes(point_t *p, const real *RHOWGDFTH, const real *SSURF, const real *SLICE, real PHIWA, real EM, real F1, int LNUPD) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real PHIOC_ICE = C_(-3.75), PHIAW_ICE = C_(3.75);
  const real C1 = C_(1.03E-3), C2 = C_(0.04E-3), P1 = C_(1.48), P2 = C_(-0.21), CDMAX_LOC = C_(0.003);
  const real EFD_MIN = C_(0.0625), EFD_MAX = C_(6.25);
  real TAU, XN, TAUO, U10P, CD_BULK, CD_WAVE, CD_ICE, EPSUS3, EPSMIN1000, ZCITHRS, CITHRSH_INV, ZMAXEXP;
  real EFD, FFD, EFD_FAC, FFD_FAC, XSTRESS, YSTRESS, XSTRESSICE, YSTRESSICE, USTAR, PHILF, OOVAL, EM_OC, F1_OC;
  real CMRHOWGDFTH, SUMT, SUMX, SUMY, SUMXICE, SUMYICE;
  EPSUS3 = S.EPSUS * SQRT(S.EPSUS);
  EPSMIN1000 = S.EPSMIN * C_(1000.0);
  if (S.c.lciwa1 || S.c.lciwa2 || S.c.lciwa3) { ZCITHRS = C_(0.); CITHRSH_INV = C_(50.); ZMAXEXP = C_(20.); }
  else { ZCITHRS = S.CIBLOCK; CITHRSH_INV = C_(1.) / RMAX(S.CITHRSH, C_(0.01)); ZMAXEXP = C_(10.); }
  EFD_FAC = C_(4.0) * S.EGRCRV / (S.G * S.G);
  FFD_FAC = POW(S.EGRCRV / S.AFCRV, C_(1.0) / S.BFCRV) * S.G;
  PHILF = C_(0.0); XSTRESS = C_(0.0); YSTRESS = C_(0.0); XSTRESSICE = C_(0.0); YSTRESSICE = C_(0.0);
  if (S.c.lwnemocouwrs) {
    for (int M = 0; M < NFRE; M++) {
      SUMXICE = S.SINTH[0] * RMIN(X3(SLICE, 0, M), -EPSMIN1000);
      SUMYICE = S.COSTH[0] * RMIN(X3(SLICE, 0, M), -EPSMIN1000);
      for (int K = 1; K < NANG; K++) {
        SUMXICE = SUMXICE + S.SINTH[K] * RMIN(X3(SLICE, K, M), -EPSMIN1000);
        SUMYICE = SUMYICE + S.COSTH[K] * RMIN(X3(SLICE, K, M), -EPSMIN1000);
      }
      XSTRESSICE = XSTRESSICE + S.ZALPWRS * SUMXICE * p->CINV[M] * S.RHOWG_DFIM[M];
      YSTRESSICE = YSTRESSICE + S.ZALPWRS * SUMYICE * p->CINV[M] * S.RHOWG_DFIM[M];
    }
  }
  for (int M = 0; M < NFRE; M++) {
    SUMT = X3(SSURF, 0, M);
    SUMX = S.SINTH[0] * X3(SSURF, 0, M);
    SUMY = S.COSTH[0] * X3(SSURF, 0, M);
    for (int K = 1; K < NANG; K++) {
      SUMT = SUMT + X3(SSURF, K, M);
      SUMX = SUMX + S.SINTH[K] * X3(SSURF, K, M);
      SUMY = SUMY + S.COSTH[K] * X3(SSURF, K, M);
    }
    PHILF = PHILF + SUMT * RHOWGDFTH[M];
    CMRHOWGDFTH = p->CINV[M] * RHOWGDFTH[M];
    XSTRESS = XSTRESS + SUMX * CMRHOWGDFTH;
    YSTRESS = YSTRESS + SUMY * CMRHOWGDFTH;
  }
  if (S.c.licerun && S.c.lwamrsetci) {
    if (p->CICOVER > ZCITHRS) {
      OOVAL = EXP(-RMIN(powi(p->CICOVER * CITHRSH_INV, 4), ZMAXEXP));
      U10P = RMAX(p->WSWAVE, S.EPSU10);
      CD_BULK = RMIN((C1 + C2 * POW(U10P, P1)) * POW(U10P, P2), CDMAX_LOC);
      CD_WAVE = (p->UFRIC / U10P) * (p->UFRIC / U10P);
      CD_ICE = OOVAL * CD_WAVE + (C_(1.0) - OOVAL) * CD_BULK;
      USTAR = RMAX(SQRT(CD_ICE) * U10P, S.EPSUS);
      EFD = RMIN(EFD_FAC * powi(USTAR, 4), EFD_MAX);
      EM_OC = RMAX(OOVAL * EM + (C_(1.0) - OOVAL) * EFD, EFD_MIN);
      FFD = FFD_FAC / USTAR;
      F1_OC = OOVAL * F1 + (C_(1.0) - OOVAL) * FFD;
      F1_OC = RMIN(RMAX(F1_OC, S.FR[1]), S.FR[NFRE - 1]);
    } else { OOVAL = C_(1.0); USTAR = p->UFRIC; EM_OC = EM; F1_OC = F1; }
  } else { OOVAL = C_(1.0); USTAR = p->UFRIC; EM_OC = EM; F1_OC = F1; }

  TAU = p->AIRD * RMAX(USTAR * USTAR, S.EPSUS);
  p->TAUXD = TAU * SIN(p->WDWAVE);
  p->TAUYD = TAU * COS(p->WDWAVE);
  p->TAUOCXD = p->TAUXD - OOVAL * XSTRESS;
  p->TAUOCYD = p->TAUYD - OOVAL * YSTRESS;
  TAUO = SQRT(p->TAUOCXD * p->TAUOCXD + p->TAUOCYD * p->TAUOCYD);
  p->TAUOC = RMIN(RMAX(TAUO / TAU, S.TAUOCMIN), S.TAUOCMAX);
  if (S.c.lwnemocouwrs) { p->TAUICX = -XSTRESSICE; p->TAUICY = -YSTRESSICE; }
  else { p->TAUICX = C_(0.0); p->TAUICY = C_(0.0); }
  if (S.c.lwcouast) {
    if (p->USTRA != C_(0.0) || p->VSTRA != C_(0.0)) {
      p->TAUXD = p->USTRA; p->TAUOCXD = p->USTRA * p->TAUOC;
      p->TAUYD = p->VSTRA; p->TAUOCYD = p->VSTRA * p->TAUOC;
    }
  }
  XN = p->AIRD * RMAX(USTAR * USTAR * USTAR, EPSUS3);
  p->PHIOCD = OOVAL * (PHILF - PHIWA) + (C_(1.0) - OOVAL) * PHIOC_ICE * XN;
  p->PHIEPS = p->PHIOCD / XN;
  p->PHIEPS = RMIN(RMAX(p->PHIEPS, S.PHIEPSMIN), S.PHIEPSMAX);
  p->PHIOCD = p->PHIEPS * XN;
  p->PHIAW = PHIWA / XN;
  p->PHIAW = OOVAL * PHIWA / XN + (C_(1.0) - OOVAL) * PHIAW_ICE;
  if (S.c.lwnemocou && LNUPD) {
    p->NPHIEPS = p->PHIEPS; p->NTAUOC = p->TAUOC;
    p->NSWH = (EM_OC != C_(0.0)) ? 4.0 * (double)SQRT(EM_OC) : 0.0; /* 4.0_JWRO*SQRT(EM_OC): the root in JWRB */
    p->NMWP = (F1_OC != C_(0.0)) ? 1.0 / (double)F1_OC : 0.0;
    if (S.c.lwnemotauoc) { p->NEMOTAUX += p->TAUOCXD; p->NEMOTAUY += p->TAUOCYD; }
    else { p->NEMOTAUX += p->TAUXD; p->NEMOTAUY += p->TAUYD; }
    p->NEMOWSWAVE += p->WSWAVE; p->NEMOPHIF += p->PHIOCD;
    p->NEMOTAUICX += p->TAUICX; p->NEMOTAUICY += p->TAUICY;
  }
}

/* imphftail.F90:73-87 */
static void imphftail(int MIJ, const real *FLM, const real *WAVNUM, const real *XK2CG, real *FL1) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real TEMP1 = C_(1.0) / XK2CG[MIJ - 1] / WAVNUM[MIJ - 1];
  for (int M = MIJ + 1; M <= NFRE; M++) {
    real TEMP2 = C_(1.0) / XK2CG[M - 1] / WAVNUM[M - 1];
    TEMP2 = TEMP2 / TEMP1;
    for (int K = 0; K < NANG; K++) {
      real TFAC = F(K, MIJ - 1);
      F(K, M - 1) = RMAX(TEMP2 * TFAC, FLM[K]);
    }
  }
}

/* setice.F90:67-86 */
static void setice(real *FL1, real CICOVER, const real *COSWDIF) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real CIREDUC, ICEFREE, TEMP;
  if (CICOVER > S.CITHRSH) { CIREDUC = RMAX(S.EPSMIN, (C_(1.0) - CICOVER)); ICEFREE = C_(0.0); }
  else { CIREDUC = C_(0.0); ICEFREE = C_(1.0); }
  TEMP = CIREDUC * S.FLMIN;
  for (int M = 0; M < NFRE; M++)
    for (int K = 0; K < NANG; K++) {
      real c = RMAX(C_(0.0), COSWDIF[K]);
      F(K, M) = F(K, M) * ICEFREE + TEMP * (c * c);
    }
}

/* stokesdrift.F90:89-142 */
static void stokesdrift(const real *FL1, const real *STOKFAC, real WSWAVE, real WDWAVE, real CICOVER, real *USTOKES,
                        real *VSTOKES) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real STMAX = C_(1.5);
  real CONST = C_(2.0) * S.DELTH * (S.ZPI * S.ZPI * S.ZPI) / S.G * powi(S.FR[S.NFRE_ODD - 1], 4);
  *USTOKES = C_(0.0); *VSTOKES = C_(0.0);
  for (int M = 0; M < S.NFRE_ODD; M++) {
    real STFAC = STOKFAC[M] * S.DFIM_SIM[M];
    for (int K = 0; K < NANG; K++) {
      real FAC3 = STFAC * F(K, M);
      *USTOKES = *USTOKES + FAC3 * S.SINTH[K];
      *VSTOKES = *VSTOKES + FAC3 * S.COSTH[K];
    }
  }
  for (int K = 0; K < NANG; K++) {
    real FAC1 = CONST * S.SINTH[K], FAC2 = CONST * S.COSTH[K];
    *USTOKES = *USTOKES + FAC1 * F(K, S.NFRE_ODD - 1);
    *VSTOKES = *VSTOKES + FAC2 * F(K, S.NFRE_ODD - 1);
  }
  if (S.c.licerun && S.c.lwamrsetci) {
    if (CICOVER > S.CITHRSH) {
      *USTOKES = C_(0.016) * WSWAVE * SIN(WDWAVE) * (C_(1.0) - CICOVER);
      *VSTOKES = C_(0.016) * WSWAVE * COS(WDWAVE) * (C_(1.0) - CICOVER);
    }
  }
  *USTOKES = RMIN(RMAX(*USTOKES, -STMAX), STMAX);
  *VSTOKES = RMIN(RMAX(*VSTOKES, -STMAX), STMAX);
}

/* implsch.F90:183-463 for one point. Returns nonzero on an abort branch. */
/* aki_ice.F90:60-112: wave number under an elastic ice sheet (Fox & Squire 1991), Newton iteration */
static real aki_ice(real G, real XK, real DEPTH, real RHOW, real CITH) {
  const real YMICE = C_(5.5E+9), RMUICE = C_(0.3), RHOI = C_(922.5), EBS = C_(0.000001), AKI_MAX = C_(20.0);
  if (CITH <= C_(0.0)) return XK;
  real FICSTF = (YMICE * powi(CITH, 3) / (12 * (1 - powi(RMUICE, 2)))) / RHOW;
  real RDH = (RHOI / RHOW) * CITH;
  real OM2 = G * XK * TANH(XK * DEPTH);
  real AKIOLD = C_(0.0);
  real AKI = RMIN(XK, POW(OM2 / RMAX(FICSTF, C_(1.0)), C_(0.2)));
  while (FABS(AKI - AKIOLD) > EBS * AKIOLD && AKI < AKI_MAX) {
    AKIOLD = AKI;
    real AKID = RMIN(DEPTH * AKI, C_(50.0));
    real Fv = FICSTF * powi(AKI, 5) + G * AKI - OM2 * (RDH * AKI + C_(1.) / TANH(AKID));
    real FPRIME = C_(5.) * FICSTF * powi(AKI, 4) + G - OM2 * (RDH - DEPTH / powi(SINH(AKID), 2));
    AKI = AKI - Fv / FPRIME;
    if (AKI <= C_(0.0)) AKI = AKI_MAX;
  }
  return AKI;
}
/* cimsstrn.F90:86-118: mean square wave strain in the sea ice */
static real cimsstrn(const real *FL1, const real *WAVNUM, real DEPTH, real CITHICK) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  const real F1LIM = S.FLMIN / S.DELTH;
  real STRN = C_(0.0);
  for (int M = 0; M < NFRE; M++) {
    real XKI = aki_ice(S.G, WAVNUM[M], DEPTH, S.ROWATER, CITHICK);
    real E = C_(0.5) * CITHICK * powi(XKI, 3) / WAVNUM[M];
    real SUME = C_(0.0);
    for (int K = 0; K < NANG; K++) SUME = SUME + F(K, M);
    if (SUME > F1LIM) STRN = STRN + powi(E, 2) * SUME * S.DFIM[M];
  }
  return STRN;
}

/* implsch.F90:288-462: everything after SNONLIN (SSOURCE of the flux diagnostics, SDIWBK, sea-ice attenuation, SBOTTOM, the limited
 * update, WNFLUXES, the second FKMEAN / FEMEANWS, IMPHFTAIL, SETICE, STOKESDRIFT).  Shared by the point-by-point path and the
 * NPROMA-blocked timing variant (ora_implsch_blk.inc). */
static void implsch_tail(real *FL1, const real *XLLWS, point_t *p, real *FLD, real *SL, real *SSOURCE, real *SLICE, const real *FLM,
                         const real *COSWDIF, const real *RHOWGDFTH, real EMEAN, real FMEAN, real F1MEAN, real FMEANWS_IN, real PHIWA) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  real DELT, DELTM, DELT5, GTEMP1, GTEMP2, FLHAB, EMEANWS, FMEANWS = FMEANWS_IN, USFM, AKMEAN, XKMEAN;
  real TEMP[NF], DELFL[NF];
  int LCFLX;
  DELT = (real)S.c.idelt;
  DELTM = C_(1.0) / DELT;
  DELT5 = (real)S.c.ximp * DELT;
  LCFLX = S.c.lwflux || S.c.lwfluxout || S.c.lwnemocou;
  if (LCFLX && S.c.lwvflx_snl)
    for (int i = 0; i < NANG * NFRE; i++) {
      GTEMP1 = RMAX((C_(1.0) - DELT5 * FLD[i]), C_(1.0));
      SSOURCE[i] = SL[i] / GTEMP1;
    }
  sdiwbk(FL1, FLD, SL, p->DEPTH, p->EMAXDPT, EMEAN, F1MEAN);
  if (S.c.licerun) { /* implsch.F90:312-339 */
    if (S.c.lciscal) {
      real BETA = C_(1.0) - p->CICOVER;
      for (int i = 0; i < NANG * NFRE; i++) { SL[i] = BETA * SL[i]; FLD[i] = BETA * FLD[i]; }
    }
    real ALPFAC = S.ZALPFACX; /* implsch.F90:195 */
    /* icebreak_modify_attenuation.F90:82-93: broken ice attenuates less */
    if (S.c.lwnemocouibr && p->IBRMEM <= S.ZIBRW_THRSH) ALPFAC = C_(1.0) / S.ZALPFACX;
    if (S.c.lciwa1) sdice1(FL1, FLD, SL, SLICE, p->CGROUP, p->CICOVER, p->CITHICK);
    if (S.c.lciwa2) sdice2(FL1, FLD, SL, SLICE, p->WAVNUM, p->CGROUP, p->CICOVER);
    if (S.c.lciwa3) sdice3(FL1, FLD, SL, SLICE, p->CGROUP, p->CICOVER, p->CITHICK, ALPFAC);
  }
  sbottom(FL1, FLD, SL, p->WAVNUM, p->DEPTH);

  /* :352-395 (LLUNSTR = F) */
  for (int M = 0; M < NFRE; M++) DELFL[M] = S.COFRM4[M] * DELT;
  USFM = p->UFRIC * RMAX(FMEANWS, FMEAN);
  for (int M = 0; M < NFRE; M++) TEMP[M] = USFM * DELFL[M];
  for (int K = 0; K < NANG; K++)
    for (int M = 0; M < NFRE; M++) {
      GTEMP1 = RMAX((C_(1.0) - DELT5 * X3(FLD, K, M)), C_(1.0));
      GTEMP2 = DELT * X3(SL, K, M) / GTEMP1;
      FLHAB = FABS(GTEMP2);
      FLHAB = RMIN(FLHAB, TEMP[M]);
      F(K, M) = F(K, M) + SIGN(FLHAB, GTEMP2);
      F(K, M) = RMAX(F(K, M), FLM[K]);
      X3(SSOURCE, K, M) = X3(SSOURCE, K, M) + DELTM * RMIN(S.FLMAX[M] - F(K, M), C_(0.0));
      F(K, M) = RMIN(F(K, M), S.FLMAX[M]);
    }
  if (LCFLX) wnfluxes(p, RHOWGDFTH, SSOURCE, SLICE, PHIWA, EMEAN, F1MEAN, 1);

  fkmean(FL1, p->WAVNUM, &EMEAN, &FMEAN, &F1MEAN, &AKMEAN, &XKMEAN);
  femeanws(FL1, XLLWS, &FMEANWS, &EMEANWS);
  imphftail(p->MIJ, FLM, p->WAVNUM, p->XK2CG, FL1);
  if (S.c.lwflux) {
    if (EMEANWS < S.WSEMEAN_MIN) { p->WSEMEAN = S.WSEMEAN_MIN; p->WSFMEAN = C_(2.) * S.FR[NFRE - 1]; }
    else { p->WSEMEAN = EMEANWS; p->WSFMEAN = FMEANWS; }
  }
  if (S.c.licerun && S.c.lmaskice) setice(FL1, p->CICOVER, COSWDIF);
  stokesdrift(FL1, p->STOKFAC, p->WSWAVE, p->WDWAVE, p->CICOVER, &p->USTOKES, &p->VSTOKES);
  if (S.c.lwnemocoustrn) p->STRNMS = cimsstrn(FL1, p->WAVNUM, p->DEPTH, p->CITHICK); /* stokestrn.F90:69-71 */
  /* stokestrn.F90:77-89: NEMO copies only when LWNEMOCOU */
  if (S.c.lwnemocou && ((S.c.lwnemocousend && S.c.lwcou) || !S.c.lwcou)) {
    if (S.c.lwnemocoustk) { p->NEMOUSTOKES = p->USTOKES; p->NEMOVSTOKES = p->VSTOKES; }
    else { p->NEMOUSTOKES = 0.0; p->NEMOVSTOKES = 0.0; }
    if (S.c.lwnemocoustrn) p->NEMOSTRN = p->STRNMS;
  }
}

static int implsch_point(real *FL1, real *XLLWS, point_t *p, real *dbg) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  static __thread real FLD[NA * NF], SL[NA * NF], SPOS[NA * NF], SSOURCE[NA * NF], SLICE[NA * NF];
  real DELT, DELTM, DELT5, GTEMP1, GTEMP2, FLHAB, RAORW, EMEAN, FMEAN, HALP = 0, EMEANWS, FMEANWS, USFM;
  real F1MEAN, AKMEAN, XKMEAN, PHIWA;
  real FLM[NA], COSWDIF[NA], SINWDIF2[NA], TEMP[NF], RHOWGDFTH[NF], DELFL[NF];
  int LCFLX;
  if (S.c.isnonlin < 0 || S.c.isnonlin > 2) return 2;
  if (S.c.lciwa1 && S.NICT == 0) return 2; /* SDICE1 needs ora_set_cideac */

  DELT = (real)S.c.idelt;
  DELTM = C_(1.0) / DELT;
  DELT5 = (real)S.c.ximp * DELT;
  LCFLX = S.c.lwflux || S.c.lwfluxout || S.c.lwnemocou;
  RAORW = RMAX(p->AIRD, C_(1.0)) * S.ROWATERM1;
  for (int K = 0; K < NANG; K++) {
    COSWDIF[K] = COS(S.TH[K] - p->WDWAVE);
    real s = SIN(S.TH[K] - p->WDWAVE);
    SINWDIF2[K] = s * s;
  }
  if (S.c.lwnemocouwrs) /* implsch.F90:205-213 (the SDICEn overwrite it when active) */
    for (int i = 0; i < NANG * NFRE; i++) SLICE[i] = C_(0.0);
  if (S.c.lbiwbk) sdepthlim(p->EMAXDPT, FL1);
  fkmean(FL1, p->WAVNUM, &EMEAN, &FMEAN, &F1MEAN, &AKMEAN, &XKMEAN);
  for (int K = 0; K < NANG; K++) {
    real c = RMAX(C_(0.0), COSWDIF[K]);
    FLM[K] = (C_(1.) - C_(0.9) * RMIN(p->CICOVER, C_(0.99))) * S.FLMIN * (c * c);
  }
  for (int ICALL = 1; ICALL <= 2; ICALL++) {
    if (sinflx(ICALL, 2, 1, FL1, p, RAORW, COSWDIF, SINWDIF2, FMEAN, &HALP, &FMEANWS, FLM, &PHIWA, FLD, SL, SPOS, RHOWGDFTH,
               XLLWS)) return 1;
  }
  if (dbg) { dbg[0] = EMEAN; dbg[1] = FMEAN; dbg[2] = F1MEAN; dbg[3] = AKMEAN; dbg[4] = XKMEAN; dbg[5] = FMEANWS; dbg[6] = PHIWA; }
  if (S.c.iphys == 0) sdissip_jan(FL1, FLD, SL, p->WAVNUM, EMEAN, F1MEAN, XKMEAN); /* sdissip.F90:76-83 */
  else sdissip_ard(FL1, FLD, SL, p->WAVNUM, p->XK2CG, p->UFRIC, COSWDIF, RAORW);
  if (LCFLX && !S.c.lwvflx_snl)
    for (int i = 0; i < NANG * NFRE; i++) SSOURCE[i] = SL[i];
  snonlin(FL1, FLD, SL, p->DEPTH, AKMEAN, p->WAVNUM);
  implsch_tail(FL1, XLLWS, p, FLD, SL, SSOURCE, SLICE, FLM, COSWDIF, RHOWGDFTH, EMEAN, FMEAN, F1MEAN, FMEANWS, PHIWA);
  return 0;
}

/*
 * Batched entry: n points.  Layouts: FL1/XLLWS [n][NANG][NFRE]; per-frequency [n][NFRE];
 * FF (forcing, inout)  [n][14]: AIRD WDWAVE CICOVER WSWAVE WSTAR USTRA VSTRA UFRIC TAUW TAUWDIR Z0M Z0B CHRNCK CITHICK
 * INTF (inout)         [n][15]: WSEMEAN WSFMEAN USTOKES VSTOKES STRNMS TAUXD TAUYD TAUOCXD TAUOCYD TAUOC TAUICX TAUICY PHIOCD PHIEPS PHIAW
 * ENV (in)             [n][2] : EMAXDPT DEPTH ; IENV [n][2]: IOBND IODP
 * MIJ out [n] (1-based); DBG optional [n][8]
 */
/* W2N (inout, may be NULL) [n][13] double: NEMOUSTOKES NEMOVSTOKES NEMOSTRN NPHIEPS NTAUOC NSWH NMWP NEMOTAUX NEMOTAUY NEMOTAUICX
 *                                          NEMOTAUICY NEMOWSWAVE NEMOPHIF (WAVE2OCEAN, yowdrvtype_config.yml) */
int ora_implsch_w2n(int n, real *FL1, const real *WAVNUM, const real *CGROUP, const real *CINV, const real *XK2CG,
                    const real *STOKFAC, const real *ENV, real *FF, real *INTF, int *MIJ, real *XLLWS, real *DBG, double *W2N);
int ora_implsch(int n, real *FL1, const real *WAVNUM, const real *CGROUP, const real *CINV, const real *XK2CG,
                const real *STOKFAC, const real *ENV, real *FF, real *INTF, int *MIJ, real *XLLWS, real *DBG) {
  return ora_implsch_w2n(n, FL1, WAVNUM, CGROUP, CINV, XK2CG, STOKFAC, ENV, FF, INTF, MIJ, XLLWS, DBG, NULL);
}
/* ENVIRONMENT%IBRMEM of the next ora_implsch* call (NULL: 1 = solid ice everywhere) */
static const real *g_ibrmem = NULL;
void ora_set_ibrmem(const real *ibrmem) { g_ibrmem = ibrmem; }

int ora_implsch_w2n(int n, real *FL1, const real *WAVNUM, const real *CGROUP, const real *CINV, const real *XK2CG,
                    const real *STOKFAC, const real *ENV, real *FF, real *INTF, int *MIJ, real *XLLWS, real *DBG, double *W2N) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  int rc = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(| : rc)
  for (int ij = 0; ij < n; ij++) {
    point_t p;
    memset(&p, 0, sizeof(p));
    p.WAVNUM = WAVNUM + (size_t)ij * NFRE; p.CGROUP = CGROUP + (size_t)ij * NFRE; p.CINV = CINV + (size_t)ij * NFRE;
    p.XK2CG = XK2CG + (size_t)ij * NFRE; p.STOKFAC = STOKFAC + (size_t)ij * NFRE;
    p.EMAXDPT = ENV[ij * 2]; p.DEPTH = ENV[ij * 2 + 1];
    p.IBRMEM = g_ibrmem ? g_ibrmem[ij] : C_(1.0);
    real *ff = FF + (size_t)ij * 14, *it = INTF + (size_t)ij * 15;
    p.AIRD = ff[0]; p.WDWAVE = ff[1]; p.CICOVER = ff[2]; p.WSWAVE = ff[3]; p.WSTAR = ff[4]; p.USTRA = ff[5]; p.VSTRA = ff[6];
    p.UFRIC = ff[7]; p.TAUW = ff[8]; p.TAUWDIR = ff[9]; p.Z0M = ff[10]; p.Z0B = ff[11]; p.CHRNCK = ff[12]; p.CITHICK = ff[13];
    p.WSEMEAN = it[0]; p.WSFMEAN = it[1]; p.USTOKES = it[2]; p.VSTOKES = it[3]; p.STRNMS = it[4]; p.TAUXD = it[5];
    p.TAUYD = it[6]; p.TAUOCXD = it[7]; p.TAUOCYD = it[8]; p.TAUOC = it[9]; p.TAUICX = it[10]; p.TAUICY = it[11];
    p.PHIOCD = it[12]; p.PHIEPS = it[13]; p.PHIAW = it[14];
    if (W2N) {
      const double *w = W2N + (size_t)ij * 13;
      p.NEMOUSTOKES = w[0]; p.NEMOVSTOKES = w[1]; p.NEMOSTRN = w[2]; p.NPHIEPS = w[3]; p.NTAUOC = w[4]; p.NSWH = w[5]; p.NMWP = w[6];
      p.NEMOTAUX = w[7]; p.NEMOTAUY = w[8]; p.NEMOTAUICX = w[9]; p.NEMOTAUICY = w[10]; p.NEMOWSWAVE = w[11]; p.NEMOPHIF = w[12];
    }
    rc |= implsch_point(FL1 + (size_t)ij * NANG * NFRE, XLLWS + (size_t)ij * NANG * NFRE, &p, DBG ? DBG + (size_t)ij * 8 : NULL);
    if (W2N) {
      double *w = W2N + (size_t)ij * 13;
      w[0] = p.NEMOUSTOKES; w[1] = p.NEMOVSTOKES; w[2] = p.NEMOSTRN; w[3] = p.NPHIEPS; w[4] = p.NTAUOC; w[5] = p.NSWH; w[6] = p.NMWP;
      w[7] = p.NEMOTAUX; w[8] = p.NEMOTAUY; w[9] = p.NEMOTAUICX; w[10] = p.NEMOTAUICY; w[11] = p.NEMOWSWAVE; w[12] = p.NEMOPHIF;
    }
    ff[0] = p.AIRD; ff[1] = p.WDWAVE; ff[2] = p.CICOVER; ff[3] = p.WSWAVE; ff[4] = p.WSTAR; ff[5] = p.USTRA; ff[6] = p.VSTRA;
    ff[7] = p.UFRIC; ff[8] = p.TAUW; ff[9] = p.TAUWDIR; ff[10] = p.Z0M; ff[11] = p.Z0B; ff[12] = p.CHRNCK; ff[13] = p.CITHICK;
    it[0] = p.WSEMEAN; it[1] = p.WSFMEAN; it[2] = p.USTOKES; it[3] = p.VSTOKES; it[4] = p.STRNMS; it[5] = p.TAUXD;
    it[6] = p.TAUYD; it[7] = p.TAUOCXD; it[8] = p.TAUOCYD; it[9] = p.TAUOC; it[10] = p.TAUICX; it[11] = p.TAUICY;
    it[12] = p.PHIOCD; it[13] = p.PHIEPS; it[14] = p.PHIAW;
    MIJ[ij] = p.MIJ;
  }
  return rc;
}

/* aki.F90:71-91 */
static real aki(real OM, real BETA) {
  const real EBS = C_(0.0001), DKMAX = C_(40.0);
  real AKM1, AKM2, AO, AKP, BO, THv, STH;
  AKM1 = OM * OM / (C_(4.0) * S.G);
  AKM2 = OM / (C_(2.0) * SQRT(S.G * BETA));
  AO = RMAX(AKM1, AKM2);
  for (;;) {
    AKP = AO;
    BO = BETA * AO;
    if (BO > DKMAX) return OM * OM / S.G;
    THv = S.G * AO * TANH(BO);
    STH = SQRT(THv);
    real ch = COSH(BO);
    AO = AO + (OM - STH) * STH * C_(2.0) / (THv / AO + S.G * BO / (ch * ch));
    if (!(FABS(AKP - AO) > EBS * AO)) return AO;
  }
}

/* depthprpt.F90:60-82 ; EMAXDPT: initdpthflds.F90:64-75 (0.0625*(GAM_B_J*DEPTH)**2) */
void ora_depthprpt(int n, const real *DEPTH, real *WAVNUM, real *CINV, real *CGROUP, real *XK2CG, real *OMOSNH2KD,
                   real *STOKFAC, real *EMAXDPT) {
  const int NFRE = S.NFRE;
  real GH = S.G / (C_(4.0) * S.PI);
  for (int ij = 0; ij < n; ij++) {
    for (int M = 0; M < NFRE; M++) {
      size_t i = (size_t)ij * NFRE + M;
      real OM = S.ZPIFR[M];
      real AK = aki(OM, DEPTH[ij]);
      WAVNUM[i] = AK;
      real AKD = AK * DEPTH[ij];
      if (AKD <= C_(10.0)) {
        CGROUP[i] = C_(0.5) * SQRT(S.G * TANH(AKD) / AK) * (C_(1.0) + C_(2.0) * AKD / SINH(C_(2.0) * AKD));
        OMOSNH2KD[i] = OM / SINH(C_(2.0) * AKD);
        STOKFAC[i] = C_(2.0) * S.G * AK * AK / (OM * TANH(C_(2.0) * AKD));
      } else {
        CGROUP[i] = GH / S.FR[M];
        OMOSNH2KD[i] = C_(0.0);
        STOKFAC[i] = C_(2.0) / S.G * OM * OM * OM;
      }
      CINV[i] = WAVNUM[i] / OM;
      XK2CG[i] = WAVNUM[i] * WAVNUM[i] * CGROUP[i];
    }
    if (EMAXDPT) { real t = S.GAM_B_J * DEPTH[ij]; EMAXDPT[ij] = C_(0.0625) * (t * t); }
  }
}

/* newwind.F90:105-161; FFN = FF_NEXT [n][14] same member order as FF */
void ora_newwind(int n, real *FF, const real *FFN) {
  real WGHT = C_(1.0) / RMAX(S.WSPMIN_RESET_TAUW, S.EPSMIN);
  const real USTMIN_RESET_TAUW = C_(0.08); /* yowwind.F90:20 */
  for (int ij = 0; ij < n; ij++) {
    real *f = FF + (size_t)ij * 14;
    const real *g = FFN + (size_t)ij * 14;
    if (S.c.icode == 3) {
      f[3] = g[3];
      if (f[3] < S.WSPMIN_RESET_TAUW) {
        real TLWMAX = WGHT * (S.ACD + S.BCD * f[3]) * (f[3] * f[3] * f[3]);
        f[8] = RMIN(f[8], TLWMAX);
      }
    } else { /* friction velocity forcing: :141-149 */
      f[7] = g[7];
      f[8] = powi(f[7], 2) * (C_(1.0) - powi(S.ALPHA / f[12], 2));
      if (f[7] < USTMIN_RESET_TAUW) f[8] = C_(0.0);
    }
    f[1] = g[1]; f[0] = g[0]; f[4] = g[4]; f[2] = g[2]; f[13] = g[13]; f[5] = g[5]; f[6] = g[6];
  }
}

/* outblock.F90:204,223-243 for the parameters 1-3 (LSECONDORDER = F: FL2ND = FL1): FEMEAN (femean.F90:84-121), STHQ
 * (sthq.F90:75-120), DOMINANT_PERIOD (dominant_period.F90:76-112, the pp1d parameter, outblock.F90:256-263).
 * OUT [n][5] = SWH, MWD (degrees, meteorological convention), MWP (or ZMISS), EM, PP1D (or ZMISS). */
void ora_outbs(int n, const real *FL1a, real ZMISS, real *OUT) {
  const int NANG = S.NANG, NFRE = S.NFRE;
#pragma omp parallel for schedule(static)
  for (int ij = 0; ij < n; ij++) {
    const real *FL1 = FL1a + (size_t)ij * NANG * NFRE;
    real EM = C_(0.0), FM = C_(0.0), TEMP2 = C_(0.0), SI = C_(0.0), CI = C_(0.0), THQ;
    for (int M = 0; M < NFRE; M++) {
      TEMP2 = RMAX(F(0, M), S.EPSMIN);
      for (int K = 1; K < NANG; K++) TEMP2 = TEMP2 + RMAX(F(K, M), S.EPSMIN);
      EM = EM + TEMP2 * S.DFIM[M];
      FM = FM + S.DFIMOFR[M] * TEMP2;
    }
    EM = EM + S.WETAIL * S.FR[NFRE - 1] * S.DELTH * TEMP2;
    FM = FM + S.FRTAIL * S.DELTH * TEMP2;
    FM = EM / FM;
    FM = RMAX(FM, S.FR[0]);
    for (int K = 0; K < NANG; K++) {
      real TEMP = C_(0.0);
      for (int M = 0; M < NFRE; M++) TEMP = TEMP + F(K, M) * S.DFIM[M];
      SI = SI + S.SINTH[K] * TEMP;
      CI = CI + S.COSTH[K] * TEMP;
    }
    if (CI == C_(0.0)) CI = S.EPSMIN;
    THQ = ATAN2(SI, CI);
    if (THQ < C_(0.0)) THQ = THQ + S.ZPI;
    /* dominant_period.F90:76-112 */
    real FCROP = C_(0.0), EM4 = C_(0.0), DP = C_(0.0);
    for (int M = 0; M < NFRE; M++)
      for (int K = 0; K < NANG; K++)
        if (F(K, M) > FCROP) FCROP = F(K, M);
    FCROP = C_(0.1) * FCROP;
    for (int M = 0; M < NFRE; M++) {
      real F1D4 = C_(0.0);
      for (int K = 0; K < NANG; K++)
        if (F(K, M) > FCROP) F1D4 = F1D4 + F(K, M) * S.DELTH;
      F1D4 = powi(F1D4, 4);
      EM4 = EM4 + S.DFIM[M] * F1D4;
      DP = DP + S.DFIMFR[M] * F1D4;
    }
    if (EM4 > C_(0.0) && DP > S.EPSMIN) DP = EM4 / DP;
    else DP = C_(0.0);
    real *o = OUT + (size_t)ij * 5;
    o[0] = C_(4.0) * SQRT(RMAX(EM, C_(0.0)));
    o[1] = FMOD(S.DEG * THQ + C_(180.0), C_(360.0));
    o[2] = (FM > C_(0.0)) ? C_(1.0) / FM : ZMISS;
    o[3] = EM;
    o[4] = (DP > C_(0.0)) ? DP : ZMISS;
  }
}

/* ---- single routines exposed for the known-answer tests (tests/test_known_answers.py) ------------------------------------ */
/* SNONLIN alone: SL, FLD start at zero */
void ora_snonlin(real *FL1, real DEPTH, real AKMEAN, const real *WAVNUM, real *SL, real *FLD) {
  for (int i = 0; i < S.NANG * S.NFRE; i++) { SL[i] = C_(0.0); FLD[i] = C_(0.0); }
  snonlin(FL1, FLD, SL, DEPTH, AKMEAN, WAVNUM);
}
/* SINPUT_ARD alone (sinput_ard.F90:153-520); aux[4] = SIG_N, TEMP2, PTURB, AIRD_PVISC as the routine formed them */
void ora_sinput_ard(int NGST, int LLSNEG, real *FL1, const real *WAVNUM, const real *CINV, const real *XK2CG, real WDWAVE, real WSWAVE,
                    real UFRIC, real Z0M, real AIRD, real WSTAR, real RNFAC, real *FLD, real *SL, real *SPOS, real *XLLWS, real *aux) {
  real COSWDIF[NA], SINWDIF2[NA];
  for (int K = 0; K < S.NANG; K++) {
    COSWDIF[K] = COS(S.TH[K] - WDWAVE);
    SINWDIF2[K] = SIN(S.TH[K] - WDWAVE) * SIN(S.TH[K] - WDWAVE);
  }
  sinput_ard(NGST, LLSNEG, FL1, WAVNUM, CINV, XK2CG, WDWAVE, WSWAVE, UFRIC, Z0M, COSWDIF, SINWDIF2, RMAX(AIRD, C_(1.0)) * S.ROWATERM1, WSTAR,
             RNFAC, FLD, SL, SPOS, XLLWS);
  for (int i = 0; i < 4; i++) aux[i] = sinput_aux[i];
}
/* SBOTTOM alone */
void ora_sbottom(real *FL1, const real *WAVNUM, real DEPTH, real *SL, real *FLD) {
  for (int i = 0; i < S.NANG * S.NFRE; i++) { SL[i] = C_(0.0); FLD[i] = C_(0.0); }
  sbottom(FL1, FLD, SL, WAVNUM, DEPTH);
}
/* SDISSIP_ARD alone */
void ora_sdissip_ard(real *FL1, const real *WAVNUM, const real *XK2CG, real UFRIC, real WDWAVE, real AIRD, real *SL, real *FLD) {
  real COSWDIF[NA];
  for (int K = 0; K < S.NANG; K++) COSWDIF[K] = COS(S.TH[K] - WDWAVE);
  for (int i = 0; i < S.NANG * S.NFRE; i++) { SL[i] = C_(0.0); FLD[i] = C_(0.0); }
  sdissip_ard(FL1, FLD, SL, WAVNUM, XK2CG, UFRIC, COSWDIF, RMAX(AIRD, C_(1.0)) * S.ROWATERM1);
}
/* AIRSEA / TAUT_Z0 alone (ICODE_WND = 3, first guess from the drag law: IUSFG = 0); out[4] = USTAR, Z0, Z0B, CHRNCK */
void ora_taut_z0(real U10, real UDIR, real TAUW, real TAUWDIR, real HALP, real RNFAC, real *out) {
  real US = C_(0.0), Z0 = C_(0.0), Z0B = C_(0.0), CH = C_(0.0);
  taut_z0(0, HALP, U10, UDIR, TAUW, TAUWDIR, RNFAC, &US, &Z0, &Z0B, &CH);
  out[0] = US; out[1] = Z0; out[2] = Z0B; out[3] = CH;
}

#include "ora_implsch_blk.inc"
